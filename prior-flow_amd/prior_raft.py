"""placeholder — replaced below once the HIP host wrapper exists"""
from .modules import state_dict_shapes  # noqa: F401
