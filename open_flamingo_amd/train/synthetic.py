"""Synthetic MMC4-/LAION-shaped batches: the tensor contract of the reference data pipeline
(open_flamingo/train/data.py:138-229 interleaved: <=256 tokens, T images; :56 LAION: 32 tokens, 1 image) without
webdataset/PIL.  ``<image>`` sits at position 0 and at every k*(L//T); ``<|endofchunk|>`` directly before each later
``<image>`` and at L-2; EOS-ish token at L-1; attention_mask all ones."""
import torch


def make_batch(B, T, L, info, device, seed=1, image_size=224, dtype=torch.float32, features_only_dim=None):
    g = torch.Generator(device="cpu").manual_seed(seed)
    vocab = info["vocab"]
    ids = torch.randint(0, vocab, (B, L), generator=g)
    step = max(L // T, 1)
    for k in range(T):
        pos = k * step
        if pos >= L:
            break
        ids[:, pos] = info["media_token_id"]
        if k > 0 and pos - 1 > 0:
            ids[:, pos - 1] = info["eoc_token_id"]
    if L >= 4:
        ids[:, L - 2] = info["eoc_token_id"]
    if features_only_dim is not None:
        vision = torch.randn(B, T, 1, 256, features_only_dim, generator=g)
    else:
        vision = torch.randn(B, T, 1, 3, image_size, image_size, generator=g)
    return dict(vision_x=vision.to(device=device, dtype=dtype), lang_x=ids.to(device),
                attention_mask=torch.ones(B, L, dtype=torch.long, device=device))


def make_labels(input_ids, media_token_id, eoc_token_id, pad_token_id):
    """Vectorised form of the per-sample Python loops in train_utils.py:127-150: ignore (-100) padding, everything
    before the first <image>, everything between an <|endofchunk|> and the next <image>, and the <image> tokens."""
    ids = input_ids
    labels = ids.clone()
    is_media = ids == media_token_id
    is_eoc = ids == eoc_token_id
    seen_media = is_media.cumsum(-1) > 0
    L = ids.shape[1]
    idx = torch.arange(L, device=ids.device).expand_as(ids)
    neg = torch.full_like(idx, -1)
    last_media = torch.where(is_media, idx, neg).cummax(-1).values             # last <image> at or before p
    last_eoc = torch.where(is_eoc, idx, neg).cummax(-1).values
    last_eoc_before = torch.cat([neg[:, :1], last_eoc[:, :-1]], dim=1)          # last <|endofchunk|> strictly before p
    # p lies between an <|endofchunk|> and the next <image> (train_utils.py:138-147)
    in_gap = last_eoc_before > last_media
    labels[~seen_media] = -100
    labels[in_gap] = -100
    labels[ids == pad_token_id] = -100
    labels[is_media] = -100
    return labels


def make_labels_laion(input_ids, media_token_id, pad_token_id):
    """LAION pass (train_utils.py:102-105): only padding and the <image> token are ignored; the trailing
    <|endofchunk|> / EOS of "<image>caption<|endofchunk|>" ARE trained on (unlike the interleaved rule above)."""
    labels = input_ids.clone()
    labels[input_ids == pad_token_id] = -100
    labels[input_ids == media_token_id] = -100
    return labels
