// Model-specialised build of the step kernel: the same source as libmjrl_hip.so's generic kernel, compiled with the
// model's sizes as constants (MJRL_SPEC_HEADER names the generated header of MJRL_SPEC_<size> defines).  Built into
// a code object per model shape (hipcc --genco) and attached to an env batch with mjrl_load_kernel().
#include <hip/hip_runtime.h>

#define MJRL_SPEC 1
#include MJRL_SPEC_HEADER
#include "mjrl_step.h"

// the sizes this code object was built for, checked against the model by mjrl_load_kernel
extern "C" __device__ const int mjrl_spec_sizes[MJRL_NSIZES] = {
#define X(name) MJRL_SPEC_##name,
    MJRL_SIZE_FIELDS(X)
#undef X
};

// layout of StepArgs, digest of the kernel sources, kernarg offset: compared with the library's by mjrl_load_kernel
extern "C" __device__ const unsigned long long mjrl_spec_abi[mj::SPEC_ABI_WORDS] = MJRL_SPEC_ABI_INIT;

// experiments (tools/build_variant.py -- -DMJRL_SPEC_WAVES_PER_EU=3): cap the registers for that many waves per SIMD
#ifdef MJRL_SPEC_WAVES_PER_EU
#define MJRL_SPEC_OCCUPANCY __attribute__((amdgpu_waves_per_eu(MJRL_SPEC_WAVES_PER_EU, MJRL_SPEC_WAVES_PER_EU)))
#else
#define MJRL_SPEC_OCCUPANCY
#endif

// Production build: no diagnostics in the kernel (mj::env_step_t<false>).  -DMJRL_DIAG (MJRL_SPEC_FLAGS, the profiling
// tools under tools/) builds the variant with the stage clock, the wave timeline, the LDS dump and -- with
// -DMJRL_STAGE_CUT -- the stage cuts, and exports mjrl_spec_diag so that mjrl_load_kernel knows which one it holds.
#ifdef MJRL_DIAG
extern "C" __device__ const int mjrl_spec_diag = 1;
#define MJRL_SPEC_DIAG true
#else
#define MJRL_SPEC_DIAG false
#endif

extern "C" __global__ __launch_bounds__(64) MJRL_SPEC_OCCUPANCY void mjrl_step_kernel_spec(const char* __restrict__ image, mj::StepArgs a) {
  extern __shared__ double lds[];
  DevModel m;
  mjrl_model_from_base(&m, (const char MJRL_GLOBAL*)image);
  // (`a` is read where the dispatch packet left it, field by field at its point of use: mj::kernarg_step_args; the
  // diagnostic build checks the offset that assumes)
  const mj::StepArgs* k = mj::kernarg_step_args(mj::STEP_ARGS_KERNARG_OFFSET);
#ifdef MJRL_DIAG
  if (k->qpos != a.qpos || k->n_env != a.n_env || k->lpt_words != a.lpt_words) __builtin_trap();
#endif
  mj::env_step_t<MJRL_SPEC_DIAG>(m, *k, lds);
}

// same signature as the step kernel plus a result word: run once by mjrl_load_kernel (mj::kernarg_selfcheck)
extern "C" __global__ __launch_bounds__(64) void mjrl_spec_selfcheck(const char* __restrict__ image, mj::StepArgs a, int* ok) {
  (void)image;
  mj::kernarg_selfcheck(a, ok);
}
