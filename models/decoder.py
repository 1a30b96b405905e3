"""`models.decoder` drop-in (reference: models/decoder.py)."""
from renderih_amd.decoder import decoder, GCN_vert_convert  # noqa: F401
from renderih_amd.model import load_decoder  # noqa: F401
