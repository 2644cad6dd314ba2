// tests/emu: type-only stand-in for <rccl/rccl.h> so that csrc/comm.hip compiles for the CPU test build.  The library itself is
// opened with dlopen by comm.hip; the CPU test build only ever uses the shared-memory double (MA_COMM=shm).  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <stddef.h>
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
