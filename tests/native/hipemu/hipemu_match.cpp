// hipemu, second library (TEST INFRASTRUCTURE ONLY): the matching path of libmvgx_hip.so compiled for the host.
// tests/_emu.py generates mvgx_match_emu.hip from openmvg_amd/csrc/mvgx_match.hip by three textual substitutions that do not
// touch the algorithm: (1) the LDS-DMA staging helpers (inline gfx950 assembly / the global_load_lds builtin) become per-lane
// memcpy calls with the same addresses, (2) the remaining `asm volatile` scheduling fences are dropped, (3) address-space
// attributes are dropped. Everything else - tile layout, MFMA fragment maps, the max3 epilogue, the DPP reductions of the
// verify stage, the batch pipeline of the host driver - is the product's code, executed under the fiber emulation.
#define HIPEMU_NO_PRODUCT
#include "hipemu.cpp"

#include <climits>

namespace {   // the kernels' `extern __shared__ char smem[]` (same unnamed namespace as the kernels below)
thread_local __attribute__((aligned(16))) char smem[160 * 1024];
}  // namespace
#include "mvgx_common.hip"
#include "mvgx_match_emu.hip"
