"""common/myhand/encoder_lijun.py of the reference."""
from renderih_amd.lijun import ResNetSimple, resnet_mid, load_encoder      # noqa: F401
