/* see ../cuda.h: test-only host emulation (nothing of cooperative groups is used by the kernels that are run) */
