// blosc2/typedefs.h -- schunk_var / schunk_var_ptr aliases (reference blosc2/typedefs.h:15-18).
#pragma once
#include "schunk.h"
