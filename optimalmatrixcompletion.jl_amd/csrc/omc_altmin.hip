// omc_altmin.hip -- alternating minimisation kernels (OMC.jl:1979-2279).  Filled in below.
#include <hip/hip_runtime.h>
#include "omc_device.h"
