"""probe_gemm with an alternative library path (experiments): AFM_LIB=/path python tools/probe_gemm_lib.py"""
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/afford-motion_amd')
from afm import ffi
if os.environ.get("AFM_LIB"):
    ffi._LIB_PATH = os.environ["AFM_LIB"]
exec(open('/root/repo/tools/probe_gemm.py').read())
