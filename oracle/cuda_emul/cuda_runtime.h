/* see cuda.h in this directory: test-only host emulation */
#include "cuda.h"
