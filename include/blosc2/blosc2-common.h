/* include/blosc2/blosc2-common.h -- present because compressed/blosc2/wrapper.h:12-14 includes it; everything the
 * reference needs from it is declared in ../blosc2.h (served by libcimg_hip.so). */
#ifndef CIMG_BLOSC2_BLOSC2_COMMON_H
#define CIMG_BLOSC2_BLOSC2_COMMON_H
#include "../blosc2.h"
#endif
