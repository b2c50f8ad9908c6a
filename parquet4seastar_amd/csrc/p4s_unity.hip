// The whole library as ONE translation unit, for builds whose device-side globals must exist once (the
// bounds-checked build, -DP4S_DEBUG_BOUNDS: its record of stray accesses is a __device__ variable that every
// kernel writes and the host reads back).  The regular build compiles the files below separately.
#include "p4s_abi.hip"
#include "p4s_abi_rle.hip"
#include "p4s_abi_levels.hip"
#include "p4s_abi_dict.hip"
#include "p4s_abi_delta.hip"
#include "p4s_abi_strings.hip"
#include "p4s_abi_misc.hip"
