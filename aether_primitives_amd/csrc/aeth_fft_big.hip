// aeth_fft_big.hip -- transforms that do not fit one workgroup's LDS:
// four-step power-of-two (N = N1 x N2) and Bluestein for arbitrary lengths.
#include "aeth_internal.h"
#include "aeth_fft_core.h"
#include "aeth_fft_plan.h"

namespace aeth {

int fft_plan_fourstep(aeth_fft *plan) { return set_error(AETH_E_UNSUPPORTED, "fourstep_pow2 not built yet (length %zu)", plan->len); }
int fft_run_fourstep(aeth_fft *plan, const float2 *, float2 *, size_t, int, float) { return set_error(AETH_E_UNSUPPORTED, "fourstep_pow2 not built yet (length %zu)", plan->len); }
int fft_plan_bluestein(aeth_fft *plan) { return set_error(AETH_E_UNSUPPORTED, "bluestein not built yet (length %zu)", plan->len); }
int fft_run_bluestein(aeth_fft *plan, const float2 *, float2 *, size_t, int, float) { return set_error(AETH_E_UNSUPPORTED, "bluestein not built yet (length %zu)", plan->len); }
void fft_plan_release_children(aeth_fft *) {}

}  // namespace aeth
