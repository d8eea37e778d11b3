#pragma once
#include "../../../calib_min.h"  // test-only stand-in, see calib_min.h
