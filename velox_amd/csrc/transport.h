// The point-to-point transport under exchange.hip: RCCL's ABI (what the exchange calls through
// its function table), and the entry points of the second provider of that table - ranks that
// share ONE GPU (RCCL refuses two ranks per device) exchange through a host shared-memory
// segment (shm_transport.hip). Selected with VX355_COMM_TRANSPORT=shm; everything above the table
// (grouped send / recv per peer, counts first, message cutting, receive slots, chunk agreement) is
// the same code either way.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace vx {

constexpr int kUniqueIdBytes = 128;
struct ncclUniqueId {
  char internal[kUniqueIdBytes];
};
using ncclComm_t = void*;
constexpr int kNcclSuccess = 0;
constexpr int kNcclInt8 = 0;  // ncclInt8 / ncclChar
constexpr int kNcclUint8 = 1;
constexpr int kNcclInt64 = 4;

namespace shmx {
int GetUniqueId(ncclUniqueId* id);
int CommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank);
int CommInitAll(ncclComm_t* comms, int world, const int* devices);
int CommDestroy(ncclComm_t comm);
int Send(const void* buf, size_t count, int dtype, int peer, ncclComm_t comm, hipStream_t stream);
int Recv(void* buf, size_t count, int dtype, int peer, ncclComm_t comm, hipStream_t stream);
int AllGather(const void* send, void* recv, size_t count, int dtype, ncclComm_t comm, hipStream_t stream);
int GroupStart();
int GroupEnd();
int CommCount(ncclComm_t comm, int* count);
int CommUserRank(ncclComm_t comm, int* rank);
int CommCuDevice(ncclComm_t comm, int* device);
const char* GetErrorString(int rc);
}  // namespace shmx

}  // namespace vx
