// rtx_shadow.hip.h -- shadow rays: light-space binning (bin / scan / scatter) and k_shadow_pass (packets with their apex at the light, affine strip test)
// Part of the one translation unit rtx_device.hip (device code, namespace rtx); not a stand-alone header.
#ifndef RTX_SHADOW_HIP_H_
#define RTX_SHADOW_HIP_H_

namespace rtx {

// exact any-hit for primitive k (wave-uniform), raytracing.cpp:569-589
__device__ __forceinline__ bool exact_any(const SceneDev& S, int k, F3 o, F3 ld, F3 nld, Exec* ex = nullptr)
{
    if (k < S.ns) {
        float4 hh = S.sph_hot[k];
        F3 td = nld;
        const int xf = S.sph_xf[k];
        ex_add(ex, xf ? EX_SPHX : EX_SPH);
        if (xf) td = normalized(inv_apply(ld, S.sph_im[k]));
        float t;
        return sphere_test(o, td, hh.x, hh.y, hh.z, hh.w, t);
    }
    TriRec T = load_tri(S.tri_hot, k - S.ns);
    const float4 e0 = S.tri_edge[(size_t)(k - S.ns) * 2], e1 = S.tri_edge[(size_t)(k - S.ns) * 2 + 1];
    float t, vd;
    int exitp = 0;
    const bool hit = triangle_test_edge(o, ld, T, e0, e1, t, vd, __builtin_inff(), ex ? &exitp : nullptr, S.mt != 0);
    if (ex) ex_tri(ex, exitp);
    return hit;
}

// Shadow rays (raytracing.cpp:560-589) are any-hit: which primitive blocks the light first does not
// matter, only whether one does.  So they are not traced in file order in one long loop.  They are
// materialised once per generation into a queue sorted by (light, direction cell) (k_shadow_bin, k_scan_*,
// k_shadow_scatter) and traced by k_shadow_pass as packets with their apex at the light: plain spheres
// and triangles through the shadow cluster hierarchy, affine spheres through a strip test of their own.
// One 32-byte record per queued ray, a = { o.xyz, ld.x }, b = { ld.y, ld.z, id (bits), affine-table key (bits) }: the scatter writes a ray to a
// slot of its own, two 16-byte stores into one 32-byte sector where seven 4-byte arrays took seven sectors (the
// scatter is bound by the number of memory requests, not by bytes); the pass reads two wide coalesced loads.
struct ShadowQ { float4 *a, *b; };

// Light-space binning.  Every shadow ray's line passes through its light, so rays towards one light
// that leave it in nearly the same direction are nearly the same line, wherever their hit points
// are.  The rays of a generation are therefore counting-sorted by (light, direction cell) -- an
// octahedral G x G map of the direction, cells in Morton order -- before the passes: a wave then
// holds 64 rays inside a cone of a degree or so with its apex AT the light, the packet test keeps a
// few dozen of the 10,000 primitives, and the 64 rays tend to be blocked by the same primitive.
// (12 bits per axis: direction maps of up to 4096 x 4096 cells; rtx_device.hip uses 256 or 512 for the sort, 256 for the affine table)
__host__ __device__ __forceinline__ unsigned morton12(unsigned x)
{
    x &= 0xfffu;
    x = (x | (x << 8)) & 0x00ff00ffu;
    x = (x | (x << 4)) & 0x0f0f0f0fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

__host__ __device__ __forceinline__ int light_cell(F3 v, int G)
{
    float s = fabsf(v.x) + fabsf(v.y) + fabsf(v.z);
    float inv = 1.0f / fmaxf(s, 1e-30f);
    float px = v.x * inv, py = v.y * inv;
    if (v.z < 0.0f) {
        float qx = (1.0f - fabsf(py)) * (px >= 0.0f ? 1.0f : -1.0f);
        float qy = (1.0f - fabsf(px)) * (py >= 0.0f ? 1.0f : -1.0f);
        px = qx; py = qy;
    }
    // fminf/fmaxf drop NaN operands, so a NaN direction lands in cell 0
    int cx = (int)fminf(fmaxf((px * 0.5f + 0.5f) * (float)G, 0.0f), (float)(G - 1));
    int cy = (int)fminf(fmaxf((py * 0.5f + 0.5f) * (float)G, 0.0f), (float)(G - 1));
    return (int)(morton12((unsigned)cx) | (morton12((unsigned)cy) << 1));
}


struct ShadowRay { F3 o, ld; int key; };

__device__ __forceinline__ ShadowRay make_shadow_ray(const SceneDev& S, const LevelDev& L, int j, int li, int G)
{
    const rtx_light& Lt = S.lights[li];
    F3 ip = f3(L.n_px[j], L.n_py[j], L.n_pz[j]);
    F3 nn = f3(L.n_nx[j], L.n_ny[j], L.n_nz[j]);
    ShadowRay r;
    r.ld = normalized(f3(Lt.point.x - ip.x, Lt.point.y - ip.y, Lt.point.z - ip.z));                  // :561-562
    r.o = f3(ip.x + nn.x * kBias, ip.y + nn.y * kBias, ip.z + nn.z * kBias);                       // :568
    r.key = li * G * G + light_cell(r.ld, G);
    return r;
}

__global__ __launch_bounds__(kBlock) void k_shadow_bin(SceneDev S, LevelDev L, const int* nnodes, int G, int* hist, int* rank)
{
    const int n = *nnodes;
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (blockIdx.x * kBlock >= n) return;  // block-uniform
    const bool live = j < n;
    const int lane = threadIdx.x & 63;
    const int jj = live ? j : 0;
    // one thread per HIT, every light in turn: the hit record is read once, not once per light
    const F3 ip = f3(L.n_px[jj], L.n_py[jj], L.n_pz[jj]);
    for (int li = 0; li < S.nl; li++) {
        int key = -1;
        if (live) {
            const rtx_light& Lt = S.lights[li];
            const F3 ldir = normalized(f3(Lt.point.x - ip.x, Lt.point.y - ip.y, Lt.point.z - ip.z));  // :561-562, as make_shadow_ray
            key = li * G * G + light_cell(ldir, G);
            L.lit[li * L.cap + j] = 1;  // until some primitive blocks it
        }
        // Neighbouring hits mostly fall into the same few cells: one returning atomic per distinct key per wave
        // instead of one per ray.  The groups are formed first (ballots only: leader, size, place in the group),
        // then every leader issues its atomic in ONE instruction and the wave waits for one round trip, where a
        // loop of atomic-then-broadcast waited for one per distinct key.
        int leader_of = lane, prefix = 0, count = 0;
        unsigned long long pending = __ballot(live);
        while (pending) {
            const int leader = __ffsll((long long)pending) - 1;
            const int k0 = __builtin_amdgcn_readlane(key, leader);
            const unsigned long long same = __ballot(live && key == k0) & pending;
            if ((same >> lane) & 1ull) {
                leader_of = leader;
                prefix = __popcll(same & ((1ull << lane) - 1ull));
                count = __popcll(same);
            }
            pending &= ~same;
        }
        int base = 0;
        if (live && leader_of == lane) base = atomicAdd(hist + key, count);
        base = __shfl(base, leader_of);
        const int myrank = base + prefix;
        if (live) rank[li * n + j] = myrank;
    }
}

// Exclusive scan of the bin counts in three small launches: per-block scan of 1024 bins, scan of the
// block sums, add-back.  nbins <= 4 Mi, so at most 4096 block sums.
// align > 1 (shadow queues): every group of 1024 bins -- a 32 x 32 square of one light's direction map --
// starts at a slot that is a multiple of `align` (64), so no wave straddles two lights or a long jump of
// the Morton curve; the slots skipped are marked as padding (id = -1) by k_scan_add.
__global__ __launch_bounds__(1024) void k_scan_blocks(const int* hist, int* offs, int nbins, int* bsum, int* braw, int align)
{
    __shared__ int part[1024];
    const int t = threadIdx.x, b = blockIdx.x * 1024 + t;
    const int v = (b < nbins) ? hist[b] : 0;
    part[t] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int x = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    if (b < nbins) offs[b] = part[t] - v;
    if (t == 1023) {
        braw[blockIdx.x] = part[1023];
        bsum[blockIdx.x] = (part[1023] + align - 1) / align * align;
    }
}

__global__ __launch_bounds__(1024) void k_scan_sums(int* bsum, int nblocks, int* qcount)
{
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (nblocks + 1023) / 1024;
    const int b0 = min(t * per, nblocks), b1 = min(b0 + per, nblocks);
    int sum = 0;
    for (int b = b0; b < b1; b++) sum += bsum[b];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int x = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int b = b0; b < b1; b++) { int v = bsum[b]; bsum[b] = run; run += v; }
    if (t == 1023) *qcount = part[1023];
}

// (256-thread workgroups: a 1024-thread one waits for a whole CU to drain when the chip is busy with the
// closest-hit kernel of the next generation, and this kernel sits between the sort and the shadow pass)
__global__ __launch_bounds__(kBlock) void k_scan_add(int* offs, int nbins, const int* bsum, const int* braw, int align, float4* pad_b)
{
    const int b = blockIdx.x * kBlock + threadIdx.x;
    const int blk = b >> 10, t = b & 1023;  // the 1024-bin group of k_scan_blocks this bin belongs to
    if (blk >= (nbins + 1023) / 1024) return;
    const int start = bsum[blk];
    if (b < nbins) offs[b] += start;
    if (pad_b) {
        const int raw = braw[blk];
        const int pad = (raw + align - 1) / align * align - raw;
        if (t < pad) pad_b[start + raw + t] = make_float4(0.0f, 0.0f, __int_as_float(-1), 0.0f);
    }
}

__global__ __launch_bounds__(kBlock) void k_shadow_scatter(SceneDev S, LevelDev L, const int* nnodes, int G, const int* offs,
                                                           const int* rank, ShadowQ Q)
{
    const int n = *nnodes;
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    // one thread per HIT, every light in turn (the hit record is read once)
    for (int li = 0; li < S.nl; li++) {
        ShadowRay r = make_shadow_ray(S, L, j, li, G);
        const int q = offs[r.key] + rank[li * n + j];
        Q.a[q] = make_float4(r.o.x, r.o.y, r.o.z, r.ld.x);
        // the ray's (light, cell) key of the affine table rides along in the record's spare word
        // (-2: the table does not cover this ray -- further from its light than the table's range, or on top of it)
        int akey = 0;
        if (S.aff_dt) {
            const rtx_light& Lt = S.lights[li];
            const float lx = Lt.point.x - r.o.x, ly = Lt.point.y - r.o.y, lz = Lt.point.z - r.o.z;
            const float dist = fsqrt(lx * lx + ly * ly + lz * lz) * 1.00001f;
            akey = (dist <= S.aff_dt[li] && dist >= 1e-2f) ? li * S.aff_G * S.aff_G + light_cell(r.ld, S.aff_G) : -2;
        }
        Q.b[q] = make_float4(r.ld.y, r.ld.z, __int_as_float(li * L.cap + j), __int_as_float(akey));
    }
}

// Affine spheres.  The reference maps only the
// DIRECTION into object space for shadow rays (:571-576); filter = line (o, q = ld*A) against the
// sphere, (1 + 2^-13)(q.X)^2 >= |q|^2 (|X|^2 - R^2)  (rtx_bounds.cpp, affine_shadow_record).
__device__ __forceinline__ float filt_affine(F3 o, F3 l, float4 m0, float4 m1, float4 m2, float4 m3)
{
    // m0 = a.x a.y a.z b.x | m1 = b.y b.z c.x c.y | m2 = c.z o.x o.y o.z | m3 = -R^2 index R ||A||_F
    float qx = __builtin_fmaf(l.x, m0.x, __builtin_fmaf(l.y, m0.w, l.z * m1.z));
    float qy = __builtin_fmaf(l.x, m0.y, __builtin_fmaf(l.y, m1.x, l.z * m1.w));
    float qz = __builtin_fmaf(l.x, m0.z, __builtin_fmaf(l.y, m1.y, l.z * m2.x));
    float X = o.x - m2.y, Y = o.y - m2.z, Z = o.z - m2.w;
    float bq = __builtin_fmaf(qx, X, __builtin_fmaf(qy, Y, qz * Z));
    float qq = __builtin_fmaf(qx, qx, __builtin_fmaf(qy, qy, qz * qz));
    float cc = __builtin_fmaf(X, X, __builtin_fmaf(Y, Y, __builtin_fmaf(Z, Z, m3.x)));
    return __builtin_fmaf(bq * (1.0f + 1.0f / 8192.0f), bq, -(qq * cc));
}

// The affine strip test (derivation in shadow_pass_packet): can the affine sphere m0..m3 pass the per-ray filter of
// ANY ray whose origin is K.Lr - D K.w + e, D in [K.Dmin, K.Dmax], |e| <= K.rperp, |K.Lr - o| <= K.Dfar, and whose
// direction is within the chord K.chord of K.w?  false = no such ray.
struct StripPacket { F3 w, Lr; float chord, Dmin, Dmax, Dfar, rperp; };

__device__ __forceinline__ bool affine_strip_pass(const StripPacket& K, float4 m0, float4 m1, float4 m2, float4 m3)
{
    const F3 w = K.w;
    // rows of A: ra = (m0.x m0.y m0.z), rb = (m0.w m1.x m1.y), rc = (m1.z m1.w m2.x); q = w A
    float qx = w.x * m0.x + w.y * m0.w + w.z * m1.z;
    float qy = w.x * m0.y + w.y * m1.x + w.z * m1.w;
    float qz = w.x * m0.z + w.y * m1.y + w.z * m2.x;
    // bounds only: v_sqrt_f32 / v_rcp_f32 (1 ulp) instead of the correctly rounded sequences, every
    // use below carries a 1e-5 .. 1e-4 relative margin
    const float qn = fsqrt(qx * qx + qy * qy + qz * qz);
    const float R = m3.z, nF = m3.w;           // sqrt(R^2) and ||A||_F, rounded up on the host; +inf: always pass
    const float qlow = qn * 0.99999f - K.chord * nF;  // <= |ld A| for every ray of the block
    if (qn < 1e18f && R < 1e18f && nF < 1e18f && qlow > 0.25f * qn && qlow > 1e-18f) {
        const float inv = frcp(qn);
        qx *= inv; qy *= inv; qz *= inv;
        // orthonormal pair around q (Duff et al.), good to ~1e-6
        const float sg = (qz >= 0.0f) ? 1.0f : -1.0f;
        const float aa = -frcp(sg + qz);
        const float bb = qx * qy * aa;
        const F3 ax1 = f3(1.0f + sg * qx * qx * aa, sg * bb, -sg * qx);
        const F3 ax2 = f3(bb, sg + qy * qy * aa, -qy);
        const F3 E = f3(m2.y - K.Lr.x, m2.z - K.Lr.y, m2.w - K.Lr.z);
        // |c - o| <= far for every ray of the block: o = Lr - D w + e
        const F3 f1 = f3(E.x + K.Dmin * w.x, E.y + K.Dmin * w.y, E.z + K.Dmin * w.z);
        const F3 f2 = f3(E.x + K.Dmax * w.x, E.y + K.Dmax * w.y, E.z + K.Dmax * w.z);
        const float far = fsqrt(fmaxf(f1.x * f1.x + f1.y * f1.y + f1.z * f1.z, f2.x * f2.x + f2.y * f2.y + f2.z * f2.z)) * 1.0001f + K.rperp;
        // a.(delta A) = delta.(A a^T): |.| <= K.chord |A a^T|; 1e-5 covers the pair's own orthogonality error
        const F3 v1 = f3(m0.x * ax1.x + m0.y * ax1.y + m0.z * ax1.z, m0.w * ax1.x + m1.x * ax1.y + m1.y * ax1.z, m1.z * ax1.x + m1.w * ax1.y + m2.x * ax1.z);
        const F3 v2 = f3(m0.x * ax2.x + m0.y * ax2.y + m0.z * ax2.z, m0.w * ax2.x + m1.x * ax2.y + m1.y * ax2.z, m1.z * ax2.x + m1.w * ax2.y + m2.x * ax2.z);
        const float iq = frcp(qlow) * 1.000001f;
        const float g1 = (K.chord * fsqrt(v1.x * v1.x + v1.y * v1.y + v1.z * v1.z) * 1.000001f + 1e-5f * nF) * iq;
        const float g2 = (K.chord * fsqrt(v2.x * v2.x + v2.y * v2.y + v2.z * v2.z) * 1.000001f + 1e-5f * nF) * iq;
        const float base_T = (R + K.rperp + 0.0112f * far) * 1.0001f + 1e-5f * (far + K.Dfar);
        const float T1 = base_T + g1 * far * 1.0001f, T2 = base_T + g2 * far * 1.0001f;
        const float t1 = ax1.x * w.x + ax1.y * w.y + ax1.z * w.z, e1 = ax1.x * E.x + ax1.y * E.y + ax1.z * E.z;
        const float t2 = ax2.x * w.x + ax2.y * w.y + ax2.z * w.z, e2 = ax2.x * E.x + ax2.y * E.y + ax2.z * E.z;
        const float lo1 = fminf(-K.Dmin * t1, -K.Dmax * t1), hi1 = fmaxf(-K.Dmin * t1, -K.Dmax * t1);
        const float lo2 = fminf(-K.Dmin * t2, -K.Dmax * t2), hi2 = fmaxf(-K.Dmin * t2, -K.Dmax * t2);
        if (e1 < lo1 - T1 || e1 > hi1 + T1 || e2 < lo2 - T2 || e2 > hi2 + T2) return false;
    }
    return true;
}

// The affine spheres a shadow ray towards light li with its direction in cell m of the G x G octahedral map
// (light_cell) can pass, as a bit mask per (light, cell): the strip test above with the CELL as the packet.
//   * directions: decode(q) = (qx, qy, 1 - |qx| - |qy|) (folded for z < 0) moves by at most sqrt(6) |dq|_max,
//     its L1 norm is 1 so its length is >= 1/sqrt(3), and normalising divides by that: every unit direction of
//     the cell is within 3 sqrt(2) / G of the centre's (half a cell is |dq|_max = 1/G); 1e-5 covers the rounding
//     of light_cell at the cell borders.
//   * origins: o = L - D w + e with 0 <= D <= |L - o| <= dt[li] (checked per wave by the pass) and
//     |e| <= |L - o| sin(angle(L - o, w)) <= dt (chord + bias / |L - o|): the line to the light leaves from the
//     hit point, the ray from a point 1e-4 off it (raytracing.cpp:568).
// Built once per context; the pass ORs the masks of the cells its rays fall into and runs the strip test of its
// own, tighter packet on those spheres only.
__host__ __device__ __forceinline__ unsigned demorton12(unsigned x)
{
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0f0f0f0fu;
    x = (x | (x >> 4)) & 0x00ff00ffu;
    x = (x | (x >> 8)) & 0x00000fffu;
    return x;
}

__global__ __launch_bounds__(kBlock) void k_affine_cells(SceneDev S, int G, const float* dt, unsigned long long* table, int words)
{
    const int cell = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);  // one wave per (light, cell)
    const int lane = threadIdx.x & 63;
    if (cell >= S.nl * G * G) return;
    const int li = cell / (G * G), m = cell % (G * G);
    const float qx = ((float)demorton12((unsigned)m) + 0.5f) / (float)G * 2.0f - 1.0f;
    const float qy = ((float)demorton12((unsigned)m >> 1) + 0.5f) / (float)G * 2.0f - 1.0f;
    F3 v;
    if (fabsf(qx) + fabsf(qy) <= 1.0f) v = f3(qx, qy, 1.0f - fabsf(qx) - fabsf(qy));
    else v = f3((1.0f - fabsf(qy)) * (qx >= 0.0f ? 1.0f : -1.0f), (1.0f - fabsf(qx)) * (qy >= 0.0f ? 1.0f : -1.0f), 1.0f - fabsf(qx) - fabsf(qy));
    const float vn = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    StripPacket K;
    K.w = f3(v.x / vn, v.y / vn, v.z / vn);
    K.Lr = f3(S.lights[li].point);
    K.chord = 4.2427f / (float)G + 1e-5f;
    const float D = dt[li];
    K.Dmin = -1e-3f;
    K.Dmax = D * 1.001f + 1e-3f;
    K.Dfar = D * 1.001f + 1e-3f;
    K.rperp = D * K.chord * 1.01f + 3e-4f;
    for (int a0 = 0; a0 < S.naff; a0 += 64) {
        const int ai = min(a0 + lane, S.naff - 1);
        const float4 m0 = S.aff_shadow[(size_t)ai * 4], m1 = S.aff_shadow[(size_t)ai * 4 + 1];
        const float4 m2 = S.aff_shadow[(size_t)ai * 4 + 2], m3 = S.aff_shadow[(size_t)ai * 4 + 3];
        const unsigned long long bal = __ballot(a0 + lane < S.naff && affine_strip_pass(K, m0, m1, m2, m3));
        if (lane == 0) table[(size_t)cell * words + (a0 >> 6)] = bal;
    }
}

// ------------------------------------------------------------------------------------------------
// Two-level member selection of the shadow pass (round 2).  A passing cluster used to cost one packet-test step over
// its 64 members, of which two in a hundred survive.  Now the passing clusters of a batch are written out as a list and
// taken EIGHT at a time: lane = (cluster j, sub-cluster s) tests the sphere around members 8s .. 8s+7 (ClusterSet::subR,
// built for the per-ray traversal of closest hit and now uploaded for the shadow set too); the passing sub-clusters are
// listed in turn and taken eight at a time, lane = (sub-cluster, member).  A sub-cluster sphere encloses its members'
// filter spheres (rtx_pack.cpp, enclose: radius from the binary32-rounded centre, rounded up), so a member some ray of the
// packet can pass lies in a sub-cluster that passes -- the argument that already covers the cluster level.
// visit(mb, k, keep, mm) sees 64 members at a time, `keep` lanes being survivors (mask mm != 0); it returns 1 when the
// wave has nothing left to trace.
// ------------------------------------------------------------------------------------------------
struct ShadowSelLds { int clist[64]; int slist[64]; };

template <typename Visit>
__device__ __forceinline__ int shadow_select(const SceneDev& S, const Packet& P, ShadowSelLds& L, const int part, const int parts,
                                             Exec* ex, int& pk_batches, int& pk_surv, Visit&& visit)
{
    const int lane = threadIdx.x & 63;
    const int j = lane >> 3;
    int out = 0;
    for (int cb = 0; cb < S.sncl && !out; cb += 64) {
        const int c = cb + lane;
        const bool mine = c < S.sncl && (c % parts) == part;
        const bool cpass = mine && packet_test(P, S.scl[min(c, S.sncl - 1)]);
        const unsigned long long cm = __ballot(cpass);
        if (mine) ex_add(ex, EX_PACKET);
        pk_batches++;
        if (cm == 0ull) continue;  // wave-uniform
        const int nc = __popcll(cm);
        if (cpass) L.clist[lanes_below(cm)] = c;
        lds_fence();
        for (int c0 = 0; c0 < nc && !out; c0 += 8) {
            const bool v1 = c0 + j < nc;
            const int sid0 = L.clist[min(c0 + j, nc - 1)] * 8 + (lane & 7);
            const bool spass = v1 && packet_test_R(P, S.ssubR[sid0]);
            if (v1) ex_add(ex, EX_PACKET);
            const unsigned long long sm = __ballot(spass);
            pk_batches++;
            if (sm == 0ull) continue;
            const int ns = __popcll(sm);
            if (spass) L.slist[lanes_below(sm)] = sid0;
            lds_fence();
            for (int s0 = 0; s0 < ns && !out; s0 += 8) {
                const bool v2 = s0 + j < ns;
                const size_t mi = (size_t)L.slist[min(s0 + j, ns - 1)] * 8 + (lane & 7);
                const float4 mb = S.smem[mi];
                const int k = v2 ? S.sidx[mi] : -1;
                const bool keep = k >= 0 && packet_test(P, mb);
                if (k >= 0) ex_add(ex, EX_PACKET);
                const unsigned long long mm = __ballot(keep);
                pk_batches++; pk_surv += __popcll(mm);
                if (mm != 0ull) out = visit(mb, k, keep, mm);
            }
            lds_fence();  // the next round of sub-clusters rewrites slist
        }
        lds_fence();  // the next batch rewrites clist
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// Pooled plain-primitive stage of the shadow pass: what generations >= 1 run (rtx_context::pool_from; generation 0's
// coherent rays keep the one-survivor-at-a-time walk, which is faster there).  Bit-identical either way; measurements in
// DESIGN.md 4.4 / 4.7.
//
// The first form of the pass took the survivors of the packet test one at a time: the member record broadcast
// by four v_readlane, the per-ray filter on SGPR operands (half the issue rate of VGPR operands on this chip,
// tools/ubench_valu2), and -- under the divergent branch of the lanes that passed -- the exact test of that ONE
// primitive on a scalar-loaded record: ten odd exact tests per wave with half the lanes live, each behind a
// scalar-load round trip, the triangles' three square roots run for a handful of lanes.  Profile (r02b): 73 %
// active lanes, SALU : VALU 0.61 (this stage, r02e: 94 %, 0.35).  Here the wave keeps LISTS in LDS and every stage runs on whatever 64 entries
// are waiting, whichever rays and primitives they belong to (the scheme of closest_pool, stages C / D):
//   P  the two-level packet test, one sphere per lane (as before)        -> candidates { bound, primitive }
//   F  eight candidates at a time, read back from LDS at a wave-uniform address (VGPR operands): each lane
//      = one ray runs the filter on all eight, then the passing (ray, primitive) pairs are pushed, candidate
//      by candidate, onto the sphere list or the triangle list
//   S  64 (ray, sphere) pairs: the exact sphere test, one pair per lane; the ray comes by ds_bpermute from
//      the lane that owns it, the 16-byte hot record by a gather (neighbouring pairs name the same sphere)
//   T  64 (ray, triangle) pairs: the exact triangle test up to the edge filter (triangle_stage1); the one in
//      ten that the areas must decide goes onto the area list with its t
//   A  64 (ray, triangle, t) entries: the three areas (triangle_stage2)
// A hit sets the ray's flag in LDS; the lanes re-read their flags after every S / T / A step and stop producing
// pairs.  Any-hit is order-free, so what is tested is a subset of what the one-at-a-time walk tests plus the
// pairs a ray had already queued when its blocker was found; the answer (some primitive blocks the ray) is the same.
// ------------------------------------------------------------------------------------------------
constexpr int kSpGroup = 8;              // candidates per filter step
constexpr int kSpCand = 64 + kSpGroup;   // fewer than a group wait, one push adds up to 64
constexpr int kSpList = 64 + 64;         // fewer than 64 wait, one push adds up to 64

struct ShadowPoolLds {
    float4 cb[kSpCand];                  // candidates: filter record { c, -R^2 }
    int ck[kSpCand];                     //             primitive
    unsigned listS[kSpList];             // ray << 26 | sphere
    unsigned listT[kSpList];             // ray << 26 | primitive index of a triangle
    unsigned listA[kSpList];             // the same, for the area stage
    float listAt[kSpList];               // ... and the t of its plane intersection
    int hit[64];                         // 1: some primitive blocks ray r
};

// idle: this lane's ray takes no part (blocked already, outside the lane block being traced, not a ray).
// Returns 1 for the lanes whose ray a plain sphere or a triangle of this part's clusters blocks.
template <bool HASEX>
__device__ __forceinline__ int shadow_plain_pool(const SceneDev& S, const Packet& Pin, ShadowPoolLds& W, ShadowSelLds& Lsel, const bool two, const F3 o, const F3 ld, const F3 nld,
                                                 const int idle, const int part, const int parts, Exec* ex,
                                                 int& pk_batches, int& pk_surv)
{
    const int lane = threadIdx.x & 63;
    constexpr unsigned kPrimMask = (1u << kPoolPrimBits) - 1u;
    const Packet& P = Pin;
    W.hit[lane] = 0;
    lds_fence();
    int nC = 0, nS = 0, nT = 0, nA = 0;
    int off = idle;  // (ints, not bools: a bool captured by reference is carried as a byte, see DESIGN.md 4.5 "compiler trap")

    // S: exact sphere tests (raytracing.cpp:202-228, :577-582) of the last cnt (<= 64) entries.  The shadow cluster set
    // holds no transformed sphere (rtx_pack.cpp: those go through the affine strip test), so the ray is (o, normalize(ld)).
    auto run_S = [&](const int want) {
        const int cnt = min(want, nS);  // (never more than the list holds: a drain at a lower threshold cannot index below the list)
        const int base = nS - cnt;
        const bool has = lane < cnt;
        const unsigned e = W.listS[base + (has ? lane : 0)];
        const int k = (int)(e & kPrimMask), rr = (int)(e >> kPoolPrimBits);
        const F3 ro = f3(__shfl(o.x, rr), __shfl(o.y, rr), __shfl(o.z, rr));
        const F3 rn = f3(__shfl(nld.x, rr), __shfl(nld.y, rr), __shfl(nld.z, rr));
        const float4 hh = S.sph_hot[k];
        if (has) ex_add(ex, EX_SPH);
        float t;
        const bool hit = sphere_test(ro, rn, hh.x, hh.y, hh.z, hh.w, t);
        if (has && hit) W.hit[rr] = 1;
        nS = base;
        lds_fence();
    };
    // A: the three areas (raytracing.cpp:352-365) for the entries the edge filter could not decide
    auto run_A = [&](const int want) {
        const int cnt = min(want, nA);  // (never more than the list holds: a drain at a lower threshold cannot index below the list)
        const int base = nA - cnt;
        const bool has = lane < cnt;
        const unsigned e = W.listA[base + (has ? lane : 0)];
        const float t = W.listAt[base + (has ? lane : 0)];
        const int k = (int)(e & kPrimMask), rr = (int)(e >> kPoolPrimBits);
        const F3 ro = f3(__shfl(o.x, rr), __shfl(o.y, rr), __shfl(o.z, rr));
        const F3 rl = f3(__shfl(ld.x, rr), __shfl(ld.y, rr), __shfl(ld.z, rr));
        const float4* p = S.tri_hot + (size_t)(k - S.ns) * 5;
        const float4 q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4];
        if (has) ex_add(ex, EX_TRI_FULL);
        const bool hit = triangle_stage2(ro, rl, t, q1, q2, q3, q4);
        if (has && hit) W.hit[rr] = 1;
        nA = base;
        lds_fence();
    };
    // T: the exact triangle test up to the edge filter (raytracing.cpp:338-350 + rtx_bounds.cpp, tri_edge_filter)
    auto run_T = [&](const int want) {
        const int cnt = min(want, nT);  // (never more than the list holds: a drain at a lower threshold cannot index below the list)
        const int base = nT - cnt;
        const bool has = lane < cnt;
        const unsigned e = W.listT[base + (has ? lane : 0)];
        const int k = (int)(e & kPrimMask), rr = (int)(e >> kPoolPrimBits);
        const F3 ro = f3(__shfl(o.x, rr), __shfl(o.y, rr), __shfl(o.z, rr));
        const F3 rl = f3(__shfl(ld.x, rr), __shfl(ld.y, rr), __shfl(ld.z, rr));
        nT = base;
        if (S.mt) {  // wave-uniform: RTX_F_TRI_MOLLER_TRUMBORE, one stage
            const TriRec T = load_tri(S.tri_hot, k - S.ns);
            float t, vd;
            if (has) ex_add(ex, EX_TRI_FULL);
            const bool hit = triangle_test_mt(ro, rl, T, t, vd) && !(t >= __builtin_inff());
            if (has && hit) W.hit[rr] = 1;
            lds_fence();
            return;
        }
        const float4* p = S.tri_hot + (size_t)(k - S.ns) * 5;
        const float4 q0 = p[0], q1 = p[1];
        const float4 e0 = S.tri_edge[(size_t)(k - S.ns) * 2], e1 = S.tri_edge[(size_t)(k - S.ns) * 2 + 1];
        float t = 0.0f;
        int exitp = 0;
        int need = triangle_stage1(ro, rl, q0, q1, e0, e1, t, HASEX ? &exitp : nullptr);
        if (!has) need = 0;
        if (HASEX && has && !need) ex_tri(ex, exitp);
        const unsigned long long bal = __ballot(need != 0);
        if (bal) {  // wave-uniform
            if (need) {
                const int pos = nA + lanes_below(bal);
                W.listA[pos] = e;
                W.listAt[pos] = t;
            }
            nA += __popcll(bal);
        }
        lds_fence();
        if (nA >= 64) run_A(64);
    };
    // F: the per-ray filter on every complete group of kSpGroup candidates, read back from LDS at wave-uniform addresses
    // (VGPR operands, eight independent chains); the passing (ray, primitive) pairs are pushed candidate by candidate
    // onto the list of their kind.  What is left over (< kSpGroup) moves to the front.  (One candidate a step, as soon
    // as a cluster's survivors are known, was measured too: fresher flags, but a ballot and a branch per candidate --
    // the generation-1 pass took 0.96 ms instead of 0.61.)
    auto run_F = [&](const int real) {  // real: candidates that are not padding (counting builds)
        int pos = 0;
        // (the filter's scaled direction is derived here rather than carried through the exact stages: registers)
        const F3 fd = f3(nld.x * kDirScale, nld.y * kDirScale, nld.z * kDirScale);
        for (; pos + kSpGroup <= nC; pos += kSpGroup) {
            unsigned pmask = 0u;
#pragma unroll
            for (int v = 0; v < kSpGroup; v++)
                if (filt_front(o, fd, W.cb[pos + v])) pmask |= 1u << v;
            if (off) pmask = 0u;
            if (HASEX && !off) ex_add(ex, EX_FILT, (unsigned)max(0, min(kSpGroup, real - pos)));
#pragma unroll
            for (int v = 0; v < kSpGroup; v++) {
                const bool pass = (pmask >> v) & 1u;
                const unsigned long long bal = __ballot(pass);
                if (bal == 0ull) continue;  // wave-uniform
                const int k = __builtin_amdgcn_readfirstlane(W.ck[pos + v]);  // (read again rather than kept: registers bound this kernel's occupancy)
                const unsigned ent = ((unsigned)lane << kPoolPrimBits) | (unsigned)k;
                if (k < S.ns) {
                    if (pass) W.listS[nS + lanes_below(bal)] = ent;
                    nS += __popcll(bal);
                    lds_fence();
                    if (nS >= 64) { run_S(64); if (W.hit[lane]) { off = 1; pmask = 0u; } }
                } else {
                    if (pass) W.listT[nT + lanes_below(bal)] = ent;
                    nT += __popcll(bal);
                    lds_fence();
                    if (nT >= 64) { run_T(64); if (W.hit[lane]) { off = 1; pmask = 0u; } }
                }
            }
        }
        const int rem = nC - pos;
        if (pos > 0 && rem > 0) {
            float4 tb = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            int tk = 0;
            if (lane < rem) { tb = W.cb[pos + lane]; tk = W.ck[pos + lane]; }
            lds_fence();
            if (lane < rem) { W.cb[lane] = tb; W.ck[lane] = tk; }
            lds_fence();
        }
        nC = rem;
    };

    // P: the packet test down the hierarchy, any order (any-hit); survivors become candidates
    bool out = __all(off != 0);
    if (two) {
      if (!out) {
        out = shadow_select(S, P, Lsel, part, parts, ex, pk_batches, pk_surv,
                            [&](const float4 mb, const int k, const bool keep, const unsigned long long mm) -> int {
            if (keep) {
                const int pos = nC + lanes_below(mm);
                W.cb[pos] = mb;
                W.ck[pos] = k;
            }
            nC += __popcll(mm);
            lds_fence();
            if (nC < kSpGroup) return 0;
            run_F(nC);
            return __all(off != 0) ? 1 : 0;
        }) != 0;
      }
    } else {
    for (int cb = 0; cb < S.sncl && !out; cb += 64) {
        const int c = cb + lane;
        unsigned long long cm = __ballot(c < S.sncl && (c % parts) == part && packet_test(P, S.scl[min(c, S.sncl - 1)]));
        if (c < S.sncl && (c % parts) == part) ex_add(ex, EX_PACKET);
        pk_batches++;
        // the members of the NEXT passing cluster are fetched while this one's are tested
        float4 mb_n = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        int k_n = -1;
        if (cm) {
            const size_t m0 = (size_t)(cb + (__ffsll((long long)cm) - 1)) * 64 + lane;
            mb_n = S.smem[m0];
            k_n = S.sidx[m0];
        }
        while (cm && !out) {
            cm &= cm - 1;
            const float4 mb = mb_n;
            const int k = k_n;
            if (cm) {
                const size_t m1 = (size_t)(cb + (__ffsll((long long)cm) - 1)) * 64 + lane;
                mb_n = S.smem[m1];
                k_n = S.sidx[m1];
            }
            const bool keep = k >= 0 && packet_test(P, mb);
            const unsigned long long mm = __ballot(keep);
            if (k >= 0) ex_add(ex, EX_PACKET);
            pk_batches++; pk_surv += __popcll(mm);
            if (mm == 0ull) continue;  // wave-uniform
            if (keep) {
                const int pos = nC + lanes_below(mm);
                W.cb[pos] = mb;
                W.ck[pos] = k;
            }
            nC += __popcll(mm);
            lds_fence();
            if (nC >= kSpGroup) {
                run_F(nC);
                out = __all(off != 0);
            }
        }
    }
    }
    // what is left: pad the last group with records no ray passes (R^2 = -inf), then drain the lists
    if (nC > 0 && !out) {
        const int real = nC;
        const int padded = (nC + kSpGroup - 1) / kSpGroup * kSpGroup;
        if (lane >= nC && lane < padded) { W.cb[lane] = make_float4(0.0f, 0.0f, 0.0f, __builtin_inff()); W.ck[lane] = S.ns; }
        nC = padded;
        lds_fence();
        run_F(real);
    }
    if (!out) {  // (all rays blocked: nothing a queued pair could change)
        if (nT > 0) run_T(nT);
        if (nS > 0) run_S(nS);
        if (nA > 0) run_A(nA);
    }
    return W.hit[lane];
}

constexpr int kAffListCap = 1024;  // affine candidates of a wave (LDS); more: every affine sphere is tested

// One shadow pass with the packet test in front of the per-ray filter (see closest_packet).
// PHASE 0: plain spheres and triangles (the shadow cluster hierarchy).  PHASE 1: affine spheres (strip test), for the
// rays PHASE 0 left lit -- a kernel of its own, launched behind PHASE 0 on the same stream when the scene has affine
// spheres: the strip test's registers (16-float records, an orthonormal frame per sphere) on top of the traversal's
// held the one-kernel form at 104 VGPRs = 4 waves per SIMD, and this kernel is bound by latency, not by issue
// (forcing 6 waves with spills: 5.59 -> 5.20 ms per 4K frame); apart, the traversal takes 65 (7 waves): 5.00 ms.
// PHASE 2: both in one kernel, the old form -- for thin generations, whose few waves do not fill the SIMDs anyway and
// which would only pay the second launch.
template <int PHASE, bool EXC, bool POOL>
__device__ __forceinline__ void shadow_pass_packet(const SceneDev& S, const ShadowQ& Qi, const int n, const int base,
                                                   unsigned char* lit, const int lit_cap, const int part, const int parts,
                                                   unsigned short* alist, ShadowPoolLds* pool, ShadowSelLds* lsel, Exec* ex)
{
    const int c0 = 0, c1 = S.nsb;
#ifdef RTX_EXPERIMENT_LAMBDA
    const bool two = false;
#else
    const bool two = S.sel2 != 0;  // which member selection this context's frames take (shadow_select; rtx_device.hip)
#endif
    const int lane = threadIdx.x & 63;
    const int i = base + threadIdx.x;
    const int ii = (i < n) ? i : 0;
    const float4 qa = Qi.a[ii], qb = Qi.b[ii];
    const int id = __float_as_int(qb.z);
    const bool live = (i < n) && id >= 0;  // id < 0: alignment padding of the bin groups
    F3 o = f3(qa.x, qa.y, qa.z);
    F3 ld = f3(qa.w, qb.x, qb.y);
    F3 nld = normalized(ld);
    const bool abn = ray_abnormal(o, ld, nld, S.guard);
    const bool use = live && !abn;
    // (the per-ray filter reads the origin itself: lanes that are done, outside the block being traced or abnormal are
    // masked by `done` / never reach a packet block, so no parked copy of the origin is kept -- registers are what bounds
    // this kernel's occupancy)
    F3 fd = use ? f3(nld.x * kDirScale, nld.y * kDirScale, nld.z * kDirScale) : f3(0.0f, 0.0f, 0.0f);
    bool done = !live;
    if (PHASE == 1) {
        // rays a plain primitive blocks were taken out of the queue by PHASE 0 (same stream): their record's id is < 0 and
        // they are not `live` here -- no gather of lit[] bytes in front of this kernel.  A split launch (several workgroups
        // per 256 rays, each clearing lit[] for its share of the clusters) cannot edit the queue it is still reading:
        if (parts != 1 && live && lit[id] == 0) done = true;
        if (__all(done)) return;                  // wave-uniform
    }
    if (PHASE != 1 && live && part == 0) ex_add(ex, EX_RAYS);
    int pk_batches = 0, pk_surv = 0, pk_blocks = 0, pk_ok = 0, pk_aff = 0;
    auto bcast4 = [](float4 v, int u) {
        float4 r;
        r.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.x), u));
        r.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.y), u));
        r.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.z), u));
        r.w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.w), u));
        return r;
    };
    // A wave is traced as ONE packet when its 64 rays make a usable one.  At a seam -- the last rays towards
    // one light followed by the first towards the next, a jump of the Morton curve over the direction map,
    // an abnormal ray -- they do not: the lanes are then split into aligned blocks of 32, 16, ... 1, each
    // traced as a packet of its own with the other lanes switched off.  A single ray that makes no packet
    // (an abnormal one) takes the transposed scan below.
    const int li = id / lit_cap;
    // The wave's affine candidates: the union of the table's masks over the (light, cell)s of its rays, as a list.
    // Any ray the table does not cover (too far from its light), or a list that does not fit: every affine sphere.
    int nlist = S.naff;
    bool listed = false;
    if (PHASE != 0 && S.aff_cells != nullptr && S.naff > 0) {
        bool covered = true;
        int key = -1;
        if (live && !abn) {  // abnormal rays take the transposed scan over everything below
            key = __float_as_int(qb.w);  // li * G * G + light_cell(ld, G), or -2: not covered (k_shadow_scatter)
            covered = key != -2;
        }
        if (__all(covered)) {
            unsigned long long acc = 0ull;
            unsigned long long pending = __ballot(key >= 0);
            while (pending) {
                const int leader = __ffsll((long long)pending) - 1;
                const int k0 = __builtin_amdgcn_readlane(key, leader);
                pending &= ~__ballot(key == k0);
                if (lane < S.aff_words) acc |= S.aff_cells[(size_t)k0 * S.aff_words + lane];
            }
            const int cnt = __popcll(acc);
            int incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(incl, off);
                if (lane >= off) incl += up;
            }
            const int total = __builtin_amdgcn_readlane(incl, 63);
            if (total <= kAffListCap) {
                int pos = incl - cnt;
                while (acc) {
                    const int b = __ffsll((long long)acc) - 1;
                    acc &= acc - 1ull;
                    alist[pos++] = (unsigned short)(lane * 64 + b);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                nlist = total;
                listed = true;
            }
        }
    }
    unsigned long long remaining = __ballot(PHASE != 1 ? live : !done), fallback = 0ull;
    int size = 64;
    while (remaining) {
        const int first = __ffsll((long long)remaining) - 1;
        const int lo = first & ~(size - 1);
        const unsigned long long blk = ((size == 64) ? ~0ull : (((1ull << size) - 1ull) << lo)) & remaining;
        const bool sel = (blk >> lane) & 1ull;
        // all of the block's lanes aimed at the same light?  then their lines share that point
        const int li0 = __builtin_amdgcn_readlane(li, first);
        const bool one_light = __all(!sel || li == li0) && li0 >= 0 && li0 < S.nl;
        F3 apex = f3(0.0f, 0.0f, 0.0f);
        if (one_light) apex = f3(S.lights[li0].point);
        const Packet P = make_packet(o, nld, sel, abn, one_light, apex);
        if (lane == 0) ex_add(ex, EX_WAVES);
        if (!P.ok && size > 1) { size >>= 1; continue; }
        remaining &= ~blk;
        if (!P.ok) { fallback |= blk; continue; }
        pk_blocks++; pk_ok += P.ok ? 1 : 0;
        // lanes outside the block sit this round out
        const bool done_keep = done;
        if (!sel) done = true;
        if (P.ok) {  // wave-uniform: two-level packet test, any order (any-hit)
            bool out = __all(done);
            if constexpr (POOL) {
            if (PHASE != 1 && !out) {
                if (shadow_plain_pool<EXC>(S, P, *pool, *lsel, two, o, ld, nld, done ? 1 : 0, part, parts, ex, pk_batches, pk_surv)) done = true;
                out = __all(done);
            }
            } else {
#if !defined(RTX_EXPERIMENT_LAMBDA)
            // the hierarchy walk of shadow_select, written out (no lambda around the per-lane bool `done`: DESIGN.md 4.5)
            const int j8 = lane >> 3;
            for (int cb = 0; two && PHASE != 1 && cb < S.sncl && !out; cb += 64) {
                const int c = cb + lane;
                const bool mine = c < S.sncl && (c % parts) == part;
                const bool cpass = mine && packet_test(P, S.scl[min(c, S.sncl - 1)]);
                const unsigned long long cm = __ballot(cpass);
                if (mine) ex_add(ex, EX_PACKET);
                pk_batches++;
                if (cm == 0ull) continue;  // wave-uniform
                const int nc = __popcll(cm);
                if (cpass) lsel->clist[lanes_below(cm)] = c;
                lds_fence();
                for (int c0 = 0; c0 < nc && !out; c0 += 8) {
                    const bool v1 = c0 + j8 < nc;
                    const int sid0 = lsel->clist[min(c0 + j8, nc - 1)] * 8 + (lane & 7);
                    const bool spass = v1 && packet_test_R(P, S.ssubR[sid0]);
                    if (v1) ex_add(ex, EX_PACKET);
                    const unsigned long long sm = __ballot(spass);
                    pk_batches++;
                    if (sm == 0ull) continue;
                    const int ns = __popcll(sm);
                    if (spass) lsel->slist[lanes_below(sm)] = sid0;
                    lds_fence();
                    for (int s0 = 0; s0 < ns && !out; s0 += 8) {
                        const bool v2 = s0 + j8 < ns;
                        const size_t mi = (size_t)lsel->slist[min(s0 + j8, ns - 1)] * 8 + (lane & 7);
                        const float4 mb = S.smem[mi];
                        const int k = v2 ? S.sidx[mi] : -1;
                        unsigned long long mm = __ballot(k >= 0 && packet_test(P, mb));
                        if (k >= 0) ex_add(ex, EX_PACKET);
                        pk_batches++; pk_surv += __popcll(mm);
                        while (mm) {
                            const int u = __ffsll((long long)mm) - 1;
                            mm &= mm - 1;
                            const float4 br = bcast4(mb, u);
                            if (!done) ex_add(ex, EX_FILT);
                            if (!done && filt_front(o, fd, br)) {
                                if (exact_any(S, __builtin_amdgcn_readlane(k, u), o, ld, nld, ex)) done = true;
                            }
                        }
                        out = __all(done);
                    }
                    lds_fence();
                }
                lds_fence();
            }
#endif
            for (int cb = 0; !two && PHASE != 1 && cb < S.sncl && !out; cb += 64) {
                const int c = cb + lane;
                unsigned long long cm = __ballot(c < S.sncl && (c % parts) == part && packet_test(P, S.scl[min(c, S.sncl - 1)]));
                if (c < S.sncl && (c % parts) == part) ex_add(ex, EX_PACKET);
                pk_batches++;
                // the members of the NEXT passing cluster are fetched while this one's survivors are traced: a wave walks
                // ~18 passing clusters, and a full memory round trip in front of each was the longest stall of its life
                float4 mb_n = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                int k_n = -1;
                if (cm && !out) {
                    const size_t m0 = (size_t)(cb + (__ffsll((long long)cm) - 1)) * 64 + lane;
                    mb_n = S.smem[m0];
                    k_n = S.sidx[m0];
                }
                while (cm && !out) {
                    cm &= cm - 1;
                    const float4 mb = mb_n;
                    const int k = k_n;
                    if (cm) {
                        const size_t m1 = (size_t)(cb + (__ffsll((long long)cm) - 1)) * 64 + lane;
                        mb_n = S.smem[m1];
                        k_n = S.sidx[m1];
                    }
                    unsigned long long mm = __ballot(k >= 0 && packet_test(P, mb));
                    if (k >= 0) ex_add(ex, EX_PACKET);
                    pk_batches++; pk_surv += __popcll(mm);
#ifdef RTX_EXPERIMENT_LAMBDA
                    // the form DESIGN.md 4.5 reports as having produced wrong lit flags in round 1 (kept out of the build)
                    auto trace_survivors = [&](const float4& mbr, const int kk, unsigned long long mask) {
                        while (mask) {
                            const int u = __ffsll((long long)mask) - 1;
                            mask &= mask - 1;
                            const float4 br = bcast4(mbr, u);
                            if (!done && filt_front(o, fd, br)) {
                                if (exact_any(S, __builtin_amdgcn_readlane(kk, u), o, ld, nld, ex)) done = true;
                            }
                        }
                    };
                    trace_survivors(mb, k, mm);
#else
                    while (mm) {
                        const int u = __ffsll((long long)mm) - 1;
                        mm &= mm - 1;
                        const float4 br = bcast4(mb, u);
                        if (!done) ex_add(ex, EX_FILT);
                        if (!done && filt_front(o, fd, br)) {
                            if (exact_any(S, __builtin_amdgcn_readlane(k, u), o, ld, nld, ex)) done = true;
                        }
                    }
#endif
                    out = __all(done);
                }
            }
            }
            // Affine spheres (raytracing.cpp:571-576: only the DIRECTION is mapped, q = ld*A, the origin stays
            // in world space).  Packet test, one sphere per lane: with a reference point Lr (the light, or the
            // packet's origin centre) every origin is Lr - D w + e, D in [Dmin, Dmax], |e| <= rperp, and every
            // direction is (w + delta) A with |delta| <= chord.  For a unit axis a orthogonal to q_w = w A the
            // point p of a line nearest the sphere centre c has a.(p - Lr) = -D (a.w) + a.e + s a.(delta A),
            // |s| |q| = |p - o| <= |c - o|, and the per-ray filter (1 + 2^-13)(q.X)^2 >= |q|^2 (|X|^2 - R^2)
            // passes only if |p - c| <= R + tan(alpha) |p - o|, tan^2(alpha) <= 2^-13 + 2^-19.  So a sphere some
            // ray can pass has, on BOTH axes of an orthonormal pair around q_w,
            //     a.(c - Lr) within [min, max](-D (a.w)) widened by T,
            //     T = R + rperp + (chord |A a^T| / min|ld A| + 0.0112) far + rounding slack,   far >= |c - o|,
            // (a.(delta A) = delta.(A a^T); |s| <= |p - o| / |ld A|).  Everything else is skipped for the whole block.
            if (PHASE != 0 && !out && S.naff > 0 && nlist > 0) {
                const F3 Lr = one_light ? apex : P.co;
                const F3 so = f3(lane_f(o.x, first), lane_f(o.y, first), lane_f(o.z, first));
                const F3 oo = f3(sel ? o.x : so.x, sel ? o.y : so.y, sel ? o.z : so.z);  // lanes outside the block borrow a ray of it
                const F3 rel = f3(Lr.x - oo.x, Lr.y - oo.y, Lr.z - oo.z);
                const float D = P.w.x * rel.x + P.w.y * rel.y + P.w.z * rel.z;
                const F3 e = f3(D * P.w.x - rel.x, D * P.w.y - rel.y, D * P.w.z - rel.z);
                const float Dmin = wave_min(D), Dmax = wave_max(D);
                const float Dfar = wave_max(fsqrt(rel.x * rel.x + rel.y * rel.y + rel.z * rel.z)) * 1.00001f;
                const float rperp = wave_max(fsqrt(e.x * e.x + e.y * e.y + e.z * e.z)) * 1.0001f + 1e-5f * Dfar;
                const float chord = fsqrt(fmaxf(0.0f, 2.0f - 2.0f * P.c0)) * 1.0001f + 2e-5f;  // |ld - w| over the block
                StripPacket K;
                K.w = P.w; K.Lr = Lr; K.chord = chord; K.Dmin = Dmin; K.Dmax = Dmax; K.Dfar = Dfar; K.rperp = rperp;
                for (int a0 = part * 64; a0 < nlist && !out; a0 += 64 * parts) {
                    const int aj = min(a0 + lane, nlist - 1);
                    const int ai = listed ? (int)alist[aj] : aj;
                    const float4 m0 = S.aff_shadow[(size_t)ai * 4], m1 = S.aff_shadow[(size_t)ai * 4 + 1];
                    const float4 m2 = S.aff_shadow[(size_t)ai * 4 + 2], m3 = S.aff_shadow[(size_t)ai * 4 + 3];
                    bool pass = a0 + lane < nlist;
                    if (pass) ex_add(ex, EX_STRIP);
                    if (pass) pass = affine_strip_pass(K, m0, m1, m2, m3);
                    unsigned long long am = __ballot(pass);
                    pk_batches++; pk_aff += __popcll(am);
                    while (am) {
                        const int u = __ffsll((long long)am) - 1;
                        am &= am - 1;
                        const float4 b0 = bcast4(m0, u), b1 = bcast4(m1, u), b2 = bcast4(m2, u), b3 = bcast4(m3, u);
                        if (!done) ex_add(ex, EX_AFILT);
                        if (!done && !(filt_affine(o, ld, b0, b1, b2, b3) < 0.0f)) {
                            if (exact_any(S, __float_as_int(b3.y), o, ld, nld, ex)) done = true;
                        }
                    }
                    out = __all(done);
                }
            }
        }
        if (!sel) done = done_keep;
    }
    // Rays that fit no packet (abnormal ones included) are traced one at a time, TRANSPOSED: the ray is
    // broadcast and the 64 lanes take 64 different primitives each round, filter and exact test.
    const int pk_fb = __popcll(fallback);
    while (fallback) {
        const int r = __ffsll((long long)fallback) - 1;
        fallback &= fallback - 1;
        auto bc = [&](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), r)); };
        const F3 ro = f3(bc(o.x), bc(o.y), bc(o.z)), rl = f3(bc(ld.x), bc(ld.y), bc(ld.z)), rn = f3(bc(nld.x), bc(nld.y), bc(nld.z));
        const F3 rfo = ro, rfd = f3(rn.x * kDirScale, rn.y * kDirScale, rn.z * kDirScale);  // (an abnormal ray bypasses the filter below)
        const bool rabn = __builtin_amdgcn_readlane((int)abn, r) != 0;
        bool blocked = false;
        for (int g = c0 + part * 64; PHASE != 1 && g < c1 && !blocked; g += 64 * parts) {
            const int gi = g + lane;
            bool hit = false;
            if (gi < c1) {
                const float4 rec = S.sbounds[gi];
                if (!rabn) ex_add(ex, EX_FILT);
                if (rabn || !(filt(rfo, rfd, rec) < 0.0f)) hit = exact_any(S, S.sorder[gi], ro, rl, rn, ex);
            }
            blocked = __any(hit);
        }
        for (int a0 = part * 64; PHASE != 0 && a0 < S.naff && !blocked; a0 += 64 * parts) {
            const int ai = a0 + lane;
            bool hit = false;
            if (ai < S.naff) {
                const float4 m0 = S.aff_shadow[(size_t)ai * 4], m1 = S.aff_shadow[(size_t)ai * 4 + 1];
                const float4 m2 = S.aff_shadow[(size_t)ai * 4 + 2], m3 = S.aff_shadow[(size_t)ai * 4 + 3];
                if (!rabn) ex_add(ex, EX_AFILT);
                if (rabn || !(filt_affine(ro, rl, m0, m1, m2, m3) < 0.0f)) hit = exact_any(S, __float_as_int(m3.y), ro, rl, rn, ex);
            }
            blocked = __any(hit);
        }
        if (lane == r && blocked) done = true;
    }
    if (PHASE != 1 && S.pstat && lane == 0) {
        atomicAdd(S.pstat + 4, (unsigned long long)pk_blocks); atomicAdd(S.pstat + 5, (unsigned long long)pk_ok + ((unsigned long long)pk_fb << 32));
        atomicAdd(S.pstat + 6, (unsigned long long)pk_batches); atomicAdd(S.pstat + 7, (unsigned long long)pk_surv + ((unsigned long long)pk_aff << 32));
    }
    if (live && done) {
        lit[id] = 0;
        // PHASE 0 with an affine kernel behind it (unsplit launches only: rtx_device.hip): the ray leaves the queue, so
        // PHASE 1 neither reads its lit byte (a scattered gather) nor traces it
        if (PHASE == 0 && S.naff > 0 && parts == 1) reinterpret_cast<int*>(Qi.b + i)[2] = -2;
    }
}

// A thin generation (<= kSplitRays shadow rays) is traced by kSplit workgroups per 256 rays, each taking every
// kSplit-th cluster and affine batch: a ray is blocked if ANY part finds a blocker (each just clears lit[id]),
// and the serial walk per wave -- what bounds the run time when there are fewer waves than SIMDs can hold --
// is kSplit times shorter.
constexpr int kSplit = 4;
constexpr int kSplitRays = 1 << 20;   // (re-measured after the edge filter: 2^21 / 2^19 cost 2 ranks 7 %, 8 ranks 4 % of their frame)
constexpr int kSplitRays8 = 1 << 17;  // ... and by 2 * kSplit workgroups below this

#ifndef RTX_WAVES_SHADOW
#define RTX_WAVES_SHADOW 4
#endif
#ifndef RTX_WAVES_POOL
#define RTX_WAVES_POOL 4
#endif
template <bool EX = false, bool POOL = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(POOL ? RTX_WAVES_POOL : RTX_WAVES_SHADOW))) void k_shadow_pass(SceneDev S, ShadowQ Qi, const int* cnt_in, unsigned char* lit, int lit_cap, int split4, int split8)
{
    const int n = *cnt_in;
    const int parts = (n <= split8) ? 2 * kSplit : (n <= split4) ? kSplit : 1;
    __shared__ ShadowPoolLds pool[POOL ? kBlock / 64 : 1];
    __shared__ ShadowSelLds lsel[kBlock / 64];
    const int base = (blockIdx.x / parts) * kBlock;
    if (base >= n) return;  // block-uniform
    Exec exs;
    Exec* ex = EX ? &exs : nullptr;
    ex_zero(ex);
    shadow_pass_packet<0, EX, POOL>(S, Qi, n, base, lit, lit_cap, blockIdx.x % parts, parts, nullptr, &pool[POOL ? (threadIdx.x >> 6) : 0],
                                    &lsel[threadIdx.x >> 6], ex);
    ex_flush(ex, S.exec + 2 * EX_KINDS);
}

// PHASE 1 of the pass: the affine spheres, for the rays still lit (launched only when the scene has affine spheres);
// BOTH = true: PHASE 2, plain primitives and affine spheres in one launch (thin generations).
template <bool EX = false, bool BOTH = false, bool POOL = false>
__global__ __launch_bounds__(kBlock) void k_shadow_affine(SceneDev S, ShadowQ Qi, const int* cnt_in, unsigned char* lit, int lit_cap, int split4, int split8)
{
    const int n = *cnt_in;
    const int parts = (n <= split8) ? 2 * kSplit : (n <= split4) ? kSplit : 1;
    __shared__ unsigned short alist[kBlock / 64][kAffListCap];
    __shared__ ShadowPoolLds pool[(BOTH && POOL) ? kBlock / 64 : 1];  // (PHASE 1 runs no plain stage)
    __shared__ ShadowSelLds lsel[BOTH ? kBlock / 64 : 1];
    const int base = (blockIdx.x / parts) * kBlock;
    if (base >= n) return;  // block-uniform
    Exec exs;
    Exec* ex = EX ? &exs : nullptr;
    ex_zero(ex);
    shadow_pass_packet<BOTH ? 2 : 1, EX, POOL>(S, Qi, n, base, lit, lit_cap, blockIdx.x % parts, parts, alist[threadIdx.x >> 6],
                                              &pool[(BOTH && POOL) ? (threadIdx.x >> 6) : 0], &lsel[BOTH ? (threadIdx.x >> 6) : 0], ex);
    ex_flush(ex, S.exec + 2 * EX_KINDS);
}


}  // namespace rtx
#endif
