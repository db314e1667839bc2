// cuda_runtime.h -- placeholder: everything lives in the force-included cuda_on_cpu.h (oracle/_ref build only)
#pragma once
#include "cuda_on_cpu.h"
