// Interface double of the handful of MPI declarations INTEGRATION.md's multi-GPU snippet uses (MPI is not in this image):
// declarations only, for a syntax check of the snippet (tests/test_docs.py).  Not an MPI implementation.
#pragma once
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
#define MPI_COMM_WORLD 0
#define MPI_DOUBLE 1
#define MPI_SUM 2
#define MPI_MAX 3
#define MPI_BYTE 4
#define MPI_SUCCESS 0
#define MPI_IN_PLACE ((void*)1)
extern "C" int MPI_Allreduce(const void* sendbuf, void* recvbuf, int count, MPI_Datatype type, MPI_Op op, MPI_Comm comm);
extern "C" int MPI_Allgather(const void* sendbuf, int sendcount, MPI_Datatype sendtype, void* recvbuf, int recvcount,
                             MPI_Datatype recvtype, MPI_Comm comm);
extern "C" int MPI_Bcast(void* buf, int count, MPI_Datatype type, int root, MPI_Comm comm);
