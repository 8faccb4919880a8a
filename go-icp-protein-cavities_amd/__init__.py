"""go-icp-protein-cavities_amd -- MI355X-native Go-ICP inner-BnB error-evaluation engine.

Python is plumbing only: this package loads the in-tree C-ABI library (csrc/ -> libgoicp_hip.so: hand-written
HIP kernels for gfx950 + the host-side GoICP mirror) through ctypes.  There is no CPU fallback: importing works
anywhere, but creating an engine without a usable MI355X raises.

The directory name is not a Python identifier; load it with
    import importlib; goicp = importlib.import_module("go-icp-protein-cavities_amd")
or `import goicp_amd` (the alias module at the repository root).
"""
from .binding import (LIB_PATH, lib, load_library, Engine, GoICP, EngineError, CFG_KEYS, DEFAULT_CFG, cfg_vector,  # noqa: F401
                      GOICP_OK, run_pair_files, run_txt_files, compute_rmsd, apply_transformation_protein, batch_search_totals, set_batch_speculation, set_batch_icp_one_launch)
from . import driver  # noqa: F401
from . import synth  # noqa: F401
from . import c4_batch  # noqa: F401
