// frad_platform.hpp -- the one seam between the kernels and the toolchain.
//
// Product build (hipcc --offload-arch=gfx950): the real HIP runtime.  The kernels are written for
// gfx950 only: wave64, LDS, no portability layer.
// Test build (g++ -DFRAD_HOST_EMULATION, tests/emu/): the same kernel source runs under a small
// thread-per-lane interpreter so that indexing and packing logic can be checked -- and run under
// AddressSanitizer, which the GPU pool does not offer -- in the CPU-only build container.  The
// emulator is test infrastructure and is never linked into libfrad_hip.so.
#pragma once
#ifdef FRAD_HOST_EMULATION
#include "hip_emu.hpp"
#else
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#define FRAD_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
// make a lane value opaque to the optimiser (used to stop loop-invariant code motion from parking
// hundreds of per-lane LDS addresses in VGPRs across a persistent loop)
// a pointer that went through a real (noinline) call is "generic" to the compiler and gets FLAT
// instructions, which also count on lgkmcnt -- an LDS-only wait would then wait for global stores.
// These casts put device-memory pointers back into the global address space.
#define FRAD_GPTR(T, p) ((__attribute__((address_space(1))) T*)(p))
#define FRAD_GCPTR(T, p) ((const __attribute__((address_space(1))) T*)(p))
#define FRAD_OPAQUE(x) asm volatile("" : "+v"(x))
// the value is complete at this point of the instruction stream: pure arithmetic otherwise floats across scheduling fences
#define FRAD_PIN(x) asm volatile("" : "+v"(x))
// workgroup barrier that only waits for this wave's LDS traffic: __syncthreads() also drains vmcnt,
// i.e. every global load and store in flight, which is exactly what a pipelined kernel must not do
#define FRAD_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
// streaming (nontemporal) 16-byte accesses for data touched once; plain accesses under the emulator
#ifdef FRAD_HOST_EMULATION
#define FRAD_NT_LOAD(p) (*(p))
#define FRAD_NT_STORE(v, p) (*(p) = (v))
#else
#define FRAD_NT_LOAD(p) __builtin_nontemporal_load(p)
#define FRAD_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif
#include <stdint.h>
