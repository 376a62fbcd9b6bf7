// sets.hip -- placeholder until the sweep kernels land
#include "internal.hpp"
using namespace sbo;
extern "C" {
int sbo_sweep_safeopt(sbo_ctx*, const sbo_sweep_opts*, sbo_safeopt_result*) { return fail(SBO_E_UNSUPPORTED, "not built yet"); }
int sbo_sweep_goose(sbo_ctx*, const sbo_sweep_opts*, sbo_goose_result*) { return fail(SBO_E_UNSUPPORTED, "not built yet"); }
int sbo_masks_get(sbo_ctx*, int, int, uint8_t*) { return fail(SBO_E_UNSUPPORTED, "not built yet"); }
}
