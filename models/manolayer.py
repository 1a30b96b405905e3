"""`models.manolayer` drop-in (reference: models/manolayer.py)."""
from renderih_amd.manolayer import ManoLayer, rodrigues_batch, vec2mat, build_mano_frame  # noqa: F401
