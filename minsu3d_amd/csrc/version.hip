#include "../../include/minsu3d_hip.h"
extern "C" const char *ms3d_version(void) { return "minsu3d_hip 0.1 (gfx950)"; }
