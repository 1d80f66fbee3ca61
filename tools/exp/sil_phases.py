"""Run `bench.py --config 5` briefly in this process, then print the rasteriser's phase stamps (10 ns ticks) if the
library was built with the temporary jrr_debug_read hook."""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
extra = sys.argv[1:]      # e.g. --batch 256
sys.argv = ['bench.py', '--config', '5', '--steps', '3', '--warmup', '1', '--no_cpu_baseline', '--no_folded', '--no_driver_blocks', '--no_rccl_one_rank', '--min_timed_ms', '100',
            '--no_skin_variants', '--no_support_tiles', '--no_config2', '--allow_experiment_lib'] + extra
import bench
bench.main()
lib = ctypes.CDLL(os.environ.get("JRR_LIB") or os.path.join(root, "joint-regressor-refinement_amd", "libjrr_hip.so"))
if hasattr(lib, 'jrr_debug_read'):
    buf = (ctypes.c_longlong * 16)(); lib.jrr_debug_read(buf)
    n = max(1, buf[7])
    names = ['set-up (vertices)', 'mesh box', 'face records + clears', 'face sweep', 'covered lists', 'resolve + adjoint atomics', 'write-out + sums']
    print('phases of workgroup 0, us per pose over', buf[7], 'poses:', {names[k]: round(buf[k] / n / 100.0, 2) for k in range(7)}, 'sum', round(sum(buf[:7]) / n / 100.0, 2))
