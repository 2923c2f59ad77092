// stub: tools/emul/hip_emul.h stands in for the HIP runtime in host emulation builds
#pragma once
#include "../../hip_emul.h"
