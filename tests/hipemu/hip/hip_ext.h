// hipemu (test infrastructure): hipExtLaunchKernelGGL lives in hip_runtime.h of the emulation.
#pragma once
#include "hip_runtime.h"
