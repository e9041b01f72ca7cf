// dispatch_host.cpp -- test shim (g++ -shared): the product's case dispatcher, csrc/mc_decide.h's mc_resolve / mc_test_face /
// mc_test_internal, compiled for the host from the very header the kernels include (through csrc/mc_device.h), with the
// lookup blob of csrc/mc_luts.h.  tests/test_dispatch_manifest.py compares every decision it takes with the manifest
// extracted mechanically from MarchingCubes.cs:94-546 (tools/gen_dispatch.py).
// The header takes its qualifiers from its includer -- <hip/hip_runtime.h> in the product; a host compiler gets them HERE:
#define __device__
#define __forceinline__ inline
#define __constant__ static const
#include "../../sdfkit_amd/csrc/mc_decide.h"

extern "C" {

// corners v[8] are iso-subtracted doubles (Cell.cs:191-208).  Returns the 8-bit sign word.
int mc_host_resolve(const double* v, int* lut_off, int* nt, int* row)
{
    const sdfk::CornersPtr c{v};
    const sdfk::Tiling t = sdfk::mc_resolve(sdfk::c_dec.v, c);
    *lut_off = t.lut_off;
    *nt = t.nt;
    *row = t.row;
    return t.index;
}

int mc_host_test_face(const double* v, int face) { return sdfk::mc_test_face(sdfk::CornersPtr{v}, face) ? 1 : 0; }

int mc_host_test_internal(const double* v, int cas, int config, int subconfig, int s)
{
    return sdfk::mc_test_internal(sdfk::c_dec.v, sdfk::CornersPtr{v}, cas, config, subconfig, s) ? 1 : 0;
}

int mc_host_interior_edge(int edge, int k) { return sdfk::c_interior_edges[edge][k]; }

}  // extern "C"
