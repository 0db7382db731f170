"""TEST-ONLY helper: build the tiny Flamingo on CPU and plug the oracle's hot-path modules into it (the product
modules refuse CPU tensors), so the host-side control flow of the boundary can be exercised without a GPU."""
import torch

from oracle import flamingo_oracle as O
from open_flamingo_amd.train import towers


def swap_in_oracle(model):
    per = O.OraclePerceiverResampler(dim=model.vis_dim, depth=len(model.perceiver.layers))
    per.load_state_dict(model.perceiver.state_dict(), strict=True)
    per.requires_grad_(True)
    model.perceiver = per
    lm = model.lang_encoder
    for i, blk in enumerate(lm.gated_cross_attn_layers):
        if blk is None:
            continue
        ob = O.OracleGatedCrossAttentionBlock(dim=model.lang_dim, dim_visual=model.vis_dim)
        ob.load_state_dict(blk.state_dict(), strict=True)
        ob.requires_grad_(True)
        lm.gated_cross_attn_layers[i] = ob
    lm.init_flamingo_layers(False)
    return model


def tiny_cpu_flamingo(seed=0, oracle=True):
    model, info = towers.build_flamingo("OF-tiny", device="cpu", seed=seed, gates=0.5)
    if oracle:
        swap_in_oracle(model)
    return model, info
