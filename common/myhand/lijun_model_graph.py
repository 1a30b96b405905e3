"""common/myhand/lijun_model_graph.py of the reference (apps/eval_interhand.py:25,238; core/graph_model.py:11,39)."""
from renderih_amd.lijun import HandNET_GCN, load_graph_model      # noqa: F401
