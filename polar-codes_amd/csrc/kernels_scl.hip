#include "polar_internal.h"
int launch_scl_decode(polar_code *, const void *, int, int, uint32_t *, double *, uint32_t *, int64_t)
{
    polar_set_error("SCL decoder not built yet");
    return POLAR_EUNSUPPORTED;
}
