// hash.cuh -- hash-to-curve on the device (placeholder during bring-up; filled in below).
#pragma once
#include "curve.cuh"
