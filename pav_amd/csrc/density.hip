// K-mer state + density scan (placeholder translation unit until the kernels land).
#include "common.h"
extern "C" void pav_density_release(pav_ctx *ctx) { (void)ctx; }
