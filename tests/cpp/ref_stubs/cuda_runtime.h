// COMPILE-ONLY stand-in: the two CUDA runtime names the reference's front-end headers mention.
#pragma once
typedef struct CUstream_st* cudaStream_t;
struct short2 { short x, y; };
struct uint3 { unsigned x, y, z; };
typedef int cudaError_t;
