/* gpu.h -- drop-in name for the reference's src/gpu/gpu.h: forwards to the HIP backend's declaration of the
 * same C API (see gpu_hip.h; every declaration cites the reference line it replaces). */
#ifndef GPU_INCLUDED
#include "gpu_hip.h"
#endif
