// oracle/_ref build only
#pragma once
#include "../cooperative_groups.h"
