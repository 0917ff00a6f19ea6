// comm.h -- what capi.hip needs of the communicator (comm.hip): the in-stream all-reduce of the
// device checkpoint's payload.  Internal to libmcmc_hip.so (hidden visibility).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

struct mcmc_hip_comm;

// ncclAllReduce(dev, dev, n doubles, op 0 = sum | 1 = max) queued on `st`; 0 or MCMC_HIP_ERR_*
int mcmc_comm_allreduce_on_stream(mcmc_hip_comm* c, double* dev, size_t n, int op, hipStream_t st);
int mcmc_comm_size(const mcmc_hip_comm* c);
int mcmc_comm_device(const mcmc_hip_comm* c);
const char* mcmc_comm_error(const mcmc_hip_comm* c);
