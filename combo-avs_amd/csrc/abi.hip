// Library introspection entry points (host only).
#include "combo_common.h"

extern "C" {
int combo_abi_version(void) { return 1; }
const char* combo_build_arch(void) { return "gfx950"; }
}
