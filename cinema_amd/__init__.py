"""cinema_amd -- MI355X-native hot path of CINeMA (conditional-INR training step and dense atlas
evaluation) behind the reference's models/inr_decoder.py + models/siren.py module API.

    from cinema_amd.models.inr_decoder import INR_Decoder
    from cinema_amd.models.siren import Siren, SineLayer
    from cinema_amd.utils import Criterion

Everything numerical runs in libcinema_hip.so (hand-written HIP for gfx950); see DESIGN.md.
"""
__version__ = "0.1.0"
