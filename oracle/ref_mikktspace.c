/* TEST INFRASTRUCTURE (oracle/_ref): a driver of ours around the reference's own tangent library, external/MikkTSpace/mikktspace.c,
 * compiled where it lies under /root/reference by oracle/Makefile.  Hands genTangSpaceDefault a triangle soup through the callbacks the
 * reference registers (src/assets/TangentGen.mm:121-179: three vertices per face, position / normal / texcoord by face corner,
 * m_setTSpaceBasic) and returns tangent + sign per corner.  Pins csrc/host/tangent_space.cpp from outside. */
#include <stddef.h>

#include "mikktspace.h"

typedef struct {
    const float* pos;
    const float* nrm;
    const float* uv;
    float* out;
    int faces;
} Soup;

static int faces_of(const SMikkTSpaceContext* c) { return ((const Soup*)c->m_pUserData)->faces; }
static int three(const SMikkTSpaceContext* c, const int f) {
    (void)c;
    (void)f;
    return 3;
}
static void position_of(const SMikkTSpaceContext* c, float out[], const int f, const int v) {
    const float* p = ((const Soup*)c->m_pUserData)->pos + 3 * ((size_t)f * 3 + (size_t)v);
    out[0] = p[0], out[1] = p[1], out[2] = p[2];
}
static void normal_of(const SMikkTSpaceContext* c, float out[], const int f, const int v) {
    const float* p = ((const Soup*)c->m_pUserData)->nrm + 3 * ((size_t)f * 3 + (size_t)v);
    out[0] = p[0], out[1] = p[1], out[2] = p[2];
}
static void texcoord_of(const SMikkTSpaceContext* c, float out[], const int f, const int v) {
    const float* p = ((const Soup*)c->m_pUserData)->uv + 2 * ((size_t)f * 3 + (size_t)v);
    out[0] = p[0], out[1] = p[1];
}
static void take(const SMikkTSpaceContext* c, const float tangent[], const float sign, const int f, const int v) {
    float* o = ((Soup*)c->m_pUserData)->out + 4 * ((size_t)f * 3 + (size_t)v);
    o[0] = tangent[0], o[1] = tangent[1], o[2] = tangent[2], o[3] = sign;
}

/* returns 1 when genTangSpaceDefault succeeded */
int ref_mikktspace(const float* positions, const float* normals, const float* uvs, int faces, float* out_tangents) {
    Soup soup = {positions, normals, uvs, out_tangents, faces};
    SMikkTSpaceInterface iface = {0};
    SMikkTSpaceContext ctx = {0};
    iface.m_getNumFaces = faces_of;
    iface.m_getNumVerticesOfFace = three;
    iface.m_getPosition = position_of;
    iface.m_getNormal = normal_of;
    iface.m_getTexCoord = texcoord_of;
    iface.m_setTSpaceBasic = take;
    ctx.m_pInterface = &iface;
    ctx.m_pUserData = &soup;
    return genTangSpaceDefault(&ctx) ? 1 : 0;
}
