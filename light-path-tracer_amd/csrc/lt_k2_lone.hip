// lt_k2_lone.hip -- the one kernel that is compiled with the ILP-first instruction scheduler.
//
// k_kerr_direct<float, Rk4<float>> carries two loops: the bulk loop, which shares its SIMD with four other waves and
// hides every result latency behind them, and the ghost-lane loop of a wavefront that is alone on its SIMD
// (lt_kernels.hpp), which is bound by the depth of its dependence chains -- a result is usable only
// ~10 cycles after its instruction issued.  LLVM's default AMDGPU scheduler orders for occupancy and lays dependent
// instructions back to back (the six-instruction reciprocal, the argument reduction of the next step's angle behind
// the last right-hand side); `-mllvm -amdgpu-sched-strategy=max-ilp` interleaves independent chains instead:
// 0.501 -> 0.484 us per step for the lone wave, the bulk loop unchanged (same instructions, 96 registers, 5 waves
// per SIMD).  The strategy is a per-translation-unit compiler option, and on every other kernel of the library it
// costs registers (prologue 38 -> 85, queue kernel 80 -> 83: an occupancy step each), hence this file: it holds the
// explicit instantiation, lt_api.hip declares it extern.  Same source, same operations: results are bit-identical
// whichever scheduler ordered them (tests/test_gpu_ghost_lanes.py digests, tools/scratch/lone_ab.sh).
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/ltrace.h"
#define LT_KERNEL_TEMPLATES_ONLY
#include "lt_kernels.hpp"

namespace lt {
template __global__ void k_kerr_direct<float, Rk4<float>>(KerrConsts<float>, const typename Vec4<float>::type *__restrict__,
                                                          typename Vec4<float>::type *__restrict__, typename Vec4<float>::type *__restrict__,
                                                          int64_t, uint32_t, uint4 *__restrict__, uint64_t *__restrict__, unsigned long long *__restrict__);
}
