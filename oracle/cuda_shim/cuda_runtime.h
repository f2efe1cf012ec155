/* see cuda.h in this directory: test-only include shim */
#include "cuda.h"
