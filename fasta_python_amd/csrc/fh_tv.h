#pragma once
// fh_tv.h -- periodic difference-stencil kernels (div / grad pair of examples/tv_denoising.py:26-63).
#include "fh_device.h"
