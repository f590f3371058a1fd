from .inr_decoder import INR_Decoder, LatentRegressor, Modulator  # noqa: F401
from .siren import SineLayer, Siren  # noqa: F401
