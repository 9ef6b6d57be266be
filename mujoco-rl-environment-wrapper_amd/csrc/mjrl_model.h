// Device-side view of the compiled model: sizes, options and one pointer per blob section
// (blob.py is the single source of the section order; mjrl_layout.h is generated from it).
// The same struct describes a host-resident blob (pointers into the blob) and, after
// mjrl_create has uploaded the sections, the copy the kernels read from HBM.
#ifndef MJRL_MODEL_H
#define MJRL_MODEL_H

#include <stddef.h>
#include <stdint.h>

#include "mjrl_layout.h"

// Section pointers always hold HBM addresses.  The device pass says so in the type (address space 1): a pointer
// loaded from the struct would otherwise be generic, and every table read a flat_load that also ties up the LDS
// counter (lgkmcnt) -- each LDS wait would then drain the table prefetches in flight.
#if defined(__HIP_DEVICE_COMPILE__)
#define MJRL_GLOBAL __attribute__((address_space(1)))
#else
#define MJRL_GLOBAL
#endif

// A model-specialised build of the step kernel (mjrl_spec_kernel.hip, one code object per model shape) defines
// MJRL_SPEC and one MJRL_SPEC_<size> per size field: there `m.nv` and friends are compile-time constants, so the LDS
// layout folds into instruction offsets, level/tree loops unroll and model-shape branches disappear.  The struct
// keeps the generic layout (the runtime copy of each size stays in place under <name>_rt).
struct DevModel {
#ifdef MJRL_SPEC
#define X(name) int name##_rt; static constexpr int name = MJRL_SPEC_##name;
#else
#define X(name) int name;
#endif
  MJRL_SIZE_FIELDS(X)
#undef X
#define X(name) double name;
  MJRL_OPT_FIELDS(X)
#undef X
#define X(name, count) const double MJRL_GLOBAL* name;
  MJRL_F64_FIELDS(X)
#undef X
#define X(name, count) const int32_t MJRL_GLOBAL* name;
  MJRL_I32_FIELDS(X)
#undef X
};

#ifdef MJRL_SPEC
// The specialised kernel builds its descriptor in registers from the image's base address alone: with the sizes
// known at compile time every section pointer is base + constant, so no descriptor field is ever loaded (the generic
// kernel pays a scalar load for the pointer and then the table read that depends on it).
__device__ __forceinline__ void mjrl_model_from_base(DevModel* m, const char MJRL_GLOBAL* b) {
  const double MJRL_GLOBAL* op = (const double MJRL_GLOBAL*)(b + 8 + 4 * MJRL_NSIZES);
  int k = 0;
#define X(name) m->name = op[k++];
  MJRL_OPT_FIELDS(X)
#undef X
  size_t off = 8 + 4 * MJRL_NSIZES + 8 * MJRL_NOPTS;
  constexpr int nq = DevModel::nq, nv = DevModel::nv, nu = DevModel::nu, nbody = DevModel::nbody, njnt = DevModel::njnt,
                ngeom = DevModel::ngeom, nsite = DevModel::nsite, ncam = DevModel::ncam, nsensor = DevModel::nsensor,
                npair = DevModel::npair, nM = DevModel::nM, ndesc = DevModel::ndesc, nchild = DevModel::nchild,
                ntree = DevModel::ntree, nfactor = DevModel::nfactor, ntab = DevModel::ntab, nchunk = DevModel::nchunk,
                ntp = DevModel::ntp, nlight = DevModel::nlight;
  (void)nchunk; (void)ntp; (void)nlight;
  (void)nfactor; (void)ntab; (void)nq; (void)nv; (void)nu; (void)nbody; (void)njnt; (void)ngeom; (void)nsite; (void)ncam; (void)nsensor;
  (void)npair; (void)nM; (void)ndesc; (void)nchild; (void)ntree;
#define X(name, count) m->name = (const double MJRL_GLOBAL*)(b + off); off += 8 * (size_t)(count);
  MJRL_F64_FIELDS(X)
#undef X
#define X(name, count) m->name = (const int32_t MJRL_GLOBAL*)(b + off); off += 4 * (size_t)(((count) + 1) & ~1);
  MJRL_I32_FIELDS(X)
#undef X
}
#else
// Point a DevModel at the sections of a blob held at `base` (host or device address; only the
// header words are dereferenced, and those come from `host_blob`).  Returns 0 on success.
static inline int mjrl_model_from_blob(DevModel* m, const void* host_blob, size_t nbytes, const void* base) {
  const char* h = (const char*)host_blob;
  if (nbytes < 8 + 4 * MJRL_NSIZES + 8 * MJRL_NOPTS) return 1;
  const int32_t* head = (const int32_t*)h;
  if (head[0] != (int32_t)MJRL_BLOB_MAGIC) return 2;
  if (head[1] != MJRL_BLOB_VERSION) return 3;
  const int32_t* sz = (const int32_t*)(h + 8);
  int k = 0;
#define X(name) m->name = sz[k++];
  MJRL_SIZE_FIELDS(X)
#undef X
  const double* op = (const double*)(h + 8 + 4 * MJRL_NSIZES);
  k = 0;
#define X(name) m->name = op[k++];
  MJRL_OPT_FIELDS(X)
#undef X
  size_t off = 8 + 4 * MJRL_NSIZES + 8 * MJRL_NOPTS;
  const char* b = (const char*)base;
  int nq = m->nq, nv = m->nv, nu = m->nu, nbody = m->nbody, njnt = m->njnt, ngeom = m->ngeom, nsite = m->nsite,
      ncam = m->ncam, nsensor = m->nsensor, npair = m->npair, nM = m->nM, ndesc = m->ndesc, nchild = m->nchild,
      ntree = m->ntree, nfactor = m->nfactor, ntab = m->ntab, nchunk = m->nchunk, ntp = m->ntp, nlight = m->nlight;
  (void)nchunk; (void)ntp; (void)nlight;
  (void)nfactor; (void)ntab; (void)nq; (void)nv; (void)nu; (void)nbody; (void)njnt; (void)ngeom; (void)nsite; (void)ncam; (void)nsensor;
  (void)npair; (void)nM; (void)ndesc; (void)nchild; (void)ntree;
#define X(name, count) m->name = (const double MJRL_GLOBAL*)(b + off); off += 8 * (size_t)(count);
  MJRL_F64_FIELDS(X)
#undef X
#define X(name, count) m->name = (const int32_t MJRL_GLOBAL*)(b + off); off += 4 * (size_t)(((count) + 1) & ~1);
  MJRL_I32_FIELDS(X)
#undef X
  return off == nbytes ? 0 : 4;
}
#endif

#endif
