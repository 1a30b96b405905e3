"""`models.encoder` drop-in (reference: models/encoder.py)."""
from renderih_amd.encoder import ResNetSimple, ResNetSimple_decoder, resnet_mid, load_encoder  # noqa: F401
