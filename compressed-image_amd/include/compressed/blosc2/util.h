// blosc2/util.h -- map_error_code (reference blosc2/util.h:16-19) is defined in wrapper.h.
#pragma once
#include "wrapper.h"
