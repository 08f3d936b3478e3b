// blosc2/lazyschunk.h -- lazy_schunk<T> lives next to schunk<T> (they share one implementation).
#pragma once
#include "schunk.h"
