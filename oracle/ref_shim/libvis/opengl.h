// libvis/opengl.h -- empty stand-in (the device-math headers include it without using anything from it); see
// ../cuda_runtime.h.  TEST INFRASTRUCTURE.
#pragma once
#include "libvis/libvis.h"
