"""common/myhand/decoder_lijun_mano.py of the reference."""
from renderih_amd.lijun import ParamRegressor, decoder_mano as decoder      # noqa: F401
