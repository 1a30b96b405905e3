"""common/myhand/decoder_lijun_graph.py of the reference."""
from renderih_amd.lijun import ParamRegressor, decoder, load_decoder      # noqa: F401
