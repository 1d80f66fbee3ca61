#!/usr/bin/env python3
"""Build tools/probe/libjrr_silstamp.so: the shipped rasteriser with wall_clock64() stamps (10 ns ticks) summed per phase over the
poses of workgroup 0, read back through jrr_debug_read (tools/exp/sil_phases.py prints them).  Phases: 0 set-up (vertices),
1 mesh box (face records + first clear are inside 2), 2 z-buffer clears, 3 face sweep, 4 resolve pass 1 (covered-pixel list), 5 resolve pass 2 (alpha, adjoint atomics),
6 write-out + reductions, 7 number of poses."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
ST = lambda k: f'  if (ADJ) {{ __syncthreads(); if (tix == 0 && blockIdx.x == 0) {{ const long long t_ = wall_clock64(); g_sil_dbg[{k}] += t_ - t_last; t_last = t_; }} }}\n'
pairs = [
    '__device__ __forceinline__ int fresh_lane() {',
    '__device__ long long g_sil_dbg[16];\n__device__ __forceinline__ int fresh_lane() {',
    '  int tix = wave_s * 64 + fresh_lane();\n',
    '  int tix = wave_s * 64 + fresh_lane();\n  long long t_last = wall_clock64();\n',
    '  __syncthreads();\n  float err = 0.f;\n',
    ST(0) + '  float err = 0.f;\n',
    '  const int bw = bx1 - bx0 + 1;\n',
    '  const int bw = bx1 - bx0 + 1;\n' + ST(1),
    '      for (int i = tix; i < npx; i += SIL_RT) zb[i] = ~0ull;\n    __syncthreads();\n',
    '      for (int i = tix; i < npx; i += SIL_RT) zb[i] = ~0ull;\n  ' + ST(2),
    '    __syncthreads();\n    // resolve, pass 1',
    '  ' + ST(3) + '    // resolve, pass 1',
    '    __syncthreads();                                                   // strip resolved before the z-buffer is reused\n',
    '  ' + ST(4),
    '  // (a FRESH laundered thread index for the write-out',
    ST(5) + '  // (a FRESH laundered thread index for the write-out',
    '      } else if (sqsil) sqsil[b] = t + (smask ? smask[b] : 0.f);\n    }\n  }\n',
    '      } else if (sqsil) sqsil[b] = t + (smask ? smask[b] : 0.f);\n    }\n  }\n  tix = tq;\n' + ST(6) + '  if (ADJ && tix == 0 && blockIdx.x == 0) g_sil_dbg[7] += 1;\n',
    '// adjoint for an arbitrary upstream gradient',
    'extern "C" void jrr_debug_read(long long* out) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sil_dbg), sizeof(long long) * 16); }\n// adjoint for an arbitrary upstream gradient',
]
sys.exit(subprocess.call([sys.executable, os.path.join(here, 'build_variant.py'), sys.argv[1] if len(sys.argv) > 1 else 'silstamp', 'sil.hip'] + sys.argv[2:] + pairs))   # extra (old, new) pairs are applied first

