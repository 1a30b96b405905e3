"""common/myhand/lijun_model_newgraph.py of the reference (core/graph_model.py:12,35; apps/eval_*.py:26)."""
from renderih_amd.lijun import HandNET_GCN, load_new_model      # noqa: F401
