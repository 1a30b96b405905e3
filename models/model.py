"""`models.model` drop-in (reference: models/model.py:18-60)."""
from renderih_amd.model import HandNET_GCN, Model, load_model, load_decoder  # noqa: F401
