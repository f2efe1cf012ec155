// testkit_poison.hip -- TEST INFRASTRUCTURE, not part of the C ABI (built into lib/libtce_testkit.so, loaded only by tests/ and scripts/probes/).
//
// tce_testkit_poison(stream): every CU's register file and LDS filled with a NaN pattern, by a launch that occupies every wave slot geometry the library's kernels use
// (256 registers per lane, 64 KiB of LDS per workgroup, two workgroups per CU, a few generations).  A kernel that reads a register or an LDS byte it never wrote -- or
// reads LDS before its DMA has landed -- normally gets what the PREVIOUS launch of the same kernel on the same data left there, i.e. the right values: such a bug passes
// every test that repeats a launch and fails in production.  Behind this launch it gets NaNs.  (Round 5: found this way -- see tests/test_gpu_w4a16_pk.py.)
#include <hip/hip_runtime.h>

namespace {

__global__ __launch_bounds__(256, 2) void poison_kernel(unsigned *sink) {
    extern __shared__ unsigned lds[];
    const unsigned pat = 0x7FC0DEADu;  // a quiet NaN as fp32; 0x7FC0 / 0xDEAD as fp16: NaN / a large negative
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = pat;
    __syncthreads();
    // all 256 vector registers: written by name, then kept alive until after the loop below by the clobber list of the second statement
#define P4(a, b, c, d) "v_mov_b32 v" #a ", %0\n\tv_mov_b32 v" #b ", %0\n\tv_mov_b32 v" #c ", %0\n\tv_mov_b32 v" #d ", %0\n\t"
    asm volatile(P4(100, 101, 102, 103) P4(104, 105, 106, 107) P4(108, 109, 110, 111) P4(112, 113, 114, 115) P4(116, 117, 118, 119) P4(120, 121, 122, 123) P4(124, 125, 126, 127)
                 P4(128, 129, 130, 131) P4(132, 133, 134, 135) P4(136, 137, 138, 139) P4(140, 141, 142, 143) P4(144, 145, 146, 147) P4(148, 149, 150, 151) P4(152, 153, 154, 155)
                 P4(156, 157, 158, 159) P4(160, 161, 162, 163) P4(164, 165, 166, 167) P4(168, 169, 170, 171) P4(172, 173, 174, 175) P4(176, 177, 178, 179) P4(180, 181, 182, 183)
                 P4(184, 185, 186, 187) P4(188, 189, 190, 191) P4(192, 193, 194, 195) P4(196, 197, 198, 199) P4(200, 201, 202, 203) P4(204, 205, 206, 207) P4(208, 209, 210, 211)
                 P4(212, 213, 214, 215) P4(216, 217, 218, 219) P4(220, 221, 222, 223) P4(224, 225, 226, 227) P4(228, 229, 230, 231) P4(232, 233, 234, 235) P4(236, 237, 238, 239)
                 P4(240, 241, 242, 243) P4(244, 245, 246, 247) P4(248, 249, 250, 251) P4(252, 253, 254, 255)
                 P4(20, 21, 22, 23) P4(24, 25, 26, 27) P4(28, 29, 30, 31) P4(32, 33, 34, 35) P4(36, 37, 38, 39) P4(40, 41, 42, 43) P4(44, 45, 46, 47) P4(48, 49, 50, 51)
                 P4(52, 53, 54, 55) P4(56, 57, 58, 59) P4(60, 61, 62, 63) P4(64, 65, 66, 67) P4(68, 69, 70, 71) P4(72, 73, 74, 75) P4(76, 77, 78, 79) P4(80, 81, 82, 83)
                 P4(84, 85, 86, 87) P4(88, 89, 90, 91) P4(92, 93, 94, 95) P4(96, 97, 98, 99)
                 ::"v"(pat)
                 : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44",
                   "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69",
                   "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94",
                   "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116",
                   "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138",
                   "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160",
                   "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182",
                   "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204",
                   "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226",
                   "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248",
                   "v249", "v250", "v251", "v252", "v253", "v254", "v255");
#undef P4
    // stay resident for a while so that the grid's generations really cover both workgroup slots of every CU
    unsigned acc = 0;
    for (int i = 0; i < 64; ++i) acc += lds[(threadIdx.x * 7 + i * 131) & 16383];
    if (acc == 0x12345u) sink[0] = acc;  // never true; keeps the loop
}

}  // namespace

extern "C" __attribute__((visibility("default"))) int tce_testkit_poison(unsigned *sink, void *stream) {
    const void *k = reinterpret_cast<const void *>(poison_kernel);
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(poison_kernel, dim3(256 * 2 * 4), dim3(256), 65536, static_cast<hipStream_t>(stream), sink);
    return (int)hipGetLastError();
}
