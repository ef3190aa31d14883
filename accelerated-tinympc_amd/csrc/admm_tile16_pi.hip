// admm_tile16_pi.hip — the PI instantiations of admm_tile16_kernel (box bounds and / or the reference per instance, fetched by LDS-DMA into
// per-wave rings: see the PI block of admm_tile16.hip) and their launcher, as a translation unit of their own.
#define TINY_T16_PI_UNIT 1
#include "admm_tile16.hip"
