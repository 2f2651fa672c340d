// Compat: third_party/RTXAdvect/cuda/DeviceTetMesh.cuh -- the device mesh handle lives in common.h.
#pragma once
#include "cuda/common.h"
