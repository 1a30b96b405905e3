"""`models.encoder` drop-in (reference: models/encoder.py)."""
from renderih_amd.encoder import (ResNetSimple, ResNetSimple_decoder, resnet_mid, HRnet_encoder, hrnet_mid,  # noqa: F401
                                  load_encoder)
