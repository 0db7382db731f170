"""MI355X-native OpenFlamingo visual-conditioning path (PerceiverResampler + GatedCrossAttentionBlock on libofhip).

Import surface mirrors ``open_flamingo/__init__.py`` of the reference:
    from open_flamingo_amd import create_model_and_transforms, Flamingo
Submodules are imported lazily so that tooling (``csrc.build``) can run before the library exists."""

__version__ = "0.1.0"


def __getattr__(name):
    if name == "create_model_and_transforms":
        from .src.factory import create_model_and_transforms
        return create_model_and_transforms
    if name == "assemble_flamingo":
        from .src.factory import assemble_flamingo
        return assemble_flamingo
    if name == "Flamingo":
        from .src.flamingo import Flamingo
        return Flamingo
    raise AttributeError(name)
