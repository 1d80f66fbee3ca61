"""Run `bench.py --config 5` briefly in this process, then print the rasteriser's phase stamps (10 ns ticks) if the
library was built with the temporary jrr_debug_read hook."""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.argv = ['bench.py', '--config', '5', '--steps', '3', '--warmup', '1', '--no_cpu_baseline', '--no_folded', '--no_driver_blocks', '--no_rccl_one_rank', '--min_timed_ms', '100']
import bench
bench.main()
lib = ctypes.CDLL(os.environ.get("JRR_LIB") or os.path.join(root, "joint-regressor-refinement_amd", "libjrr_hip.so"))
if hasattr(lib, 'jrr_debug_read'):
    buf = (ctypes.c_longlong * 16)(); lib.jrr_debug_read(buf); print('phases:', list(buf))
