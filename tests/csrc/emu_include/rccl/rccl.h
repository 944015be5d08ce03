/* Test harness (not shipped): the handful of RCCL types fk_shard.hip names, for the CPU build of the library under
   tests/csrc/hip_emu.h (the library binds RCCL with dlopen at run time; the emulated build never gets that far: one rank,
   no collectives).  Not RCCL's header and not a substitute for it: the product is compiled against /opt/rocm's. */
#pragma once
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
