// macros.h -- kept for source compatibility; everything lives in detail/config.h
#pragma once
#include "detail/config.h"
