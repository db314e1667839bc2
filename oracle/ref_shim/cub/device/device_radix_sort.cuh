// oracle/_ref build only
#pragma once
#include "../cub.cuh"
