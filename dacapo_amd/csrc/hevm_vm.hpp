// HEVM virtual machine on the MI355X: state and host-side helpers behind the 18-symbol ABI of
// include/hevm_abi.h.  Mirrors struct SEAL_HEVM (/root/reference/lib/Runtime/SEAL_HEVM.cpp:15-402) with the
// SEAL objects replaced by HBM-resident limb arrays and HIP launches.
#pragma once
#include "wire_parse.hpp"
#include <complex>
#include <deque>
#include <map>
#include <set>
#include <memory>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/hevm_abi.h"
#include "chacha.hpp"
#include "seal_serial.hpp"
#include "kernels.hpp"
#include "encoder.hpp"
#include "plan.hpp"

namespace dacapo {

// Device allocations made on behalf of one VM, so that hevm_destroy (an extension: the reference's ABI never frees a VM) can return
// them.  The C ABI's entry points name the VM they act on (HEVM::enter); allocation sites call vm_malloc / vm_free instead of
// hipMalloc / hipFree.  Allocations made with no VM entered are not tracked (and live as long as the process, as before).
struct VmAllocs {
    std::unordered_set<void *> live;
};
extern thread_local VmAllocs *g_vm_allocs;
template <class T>
inline hipError_t vm_malloc(T **p, size_t bytes)
{
    const hipError_t e = hipMalloc((void **)p, bytes);
    if (e == hipSuccess && g_vm_allocs) g_vm_allocs->live.insert((void *)*p);
    return e;
}
inline hipError_t vm_free(void *p)
{
    if (p && g_vm_allocs) g_vm_allocs->live.erase(p);
    return hipFree(p);
}


// the wire structs (WireHeader, WireConfigBody, WireOp) and the two files' parsers live in a host-only TU: wire_parse.hpp

// CKKSEncoder restated on the host (encode/decode are untimed set-up work in the reference: SEAL_HEVM.cpp:242-267,
// :439-455); the RNS lift and every NTT run on the GPU.
class HostEncoder {
  public:
    HostEncoder(int logN);
    // values tiled over N/2 slots as src[i % len] (SEAL_HEVM.cpp:259-261); returns round(coefficients) as int128
    void encode(const double *src, size_t len, double scale, std::vector<__int128> &coeffs) const;
    // coeffs: N real coefficients already divided by the scale -> N/2 slot values (real parts)
    void decode(std::vector<std::complex<double>> &coeffs, double *out) const;
    size_t N, slots;
    int logN;
    const std::vector<std::complex<double>> &roots() const { return root_; }
    const std::vector<uint32_t> &slot_map() const { return slot_map_; }

  private:
    std::vector<std::complex<double>> root_; // exp(2 pi i bitrev(k) / 2N)
    std::vector<uint32_t> slot_map_;         // CKKSEncoder::matrix_reps_index_map_
};

struct Plain {
    u64 *d = nullptr;
    int level = 0;
    double scale = 1.0;
    bool arena = false; // d points into HEVM::plain_arenas (batched device encoder): not freed individually
    u64 *dsp = nullptr; // option hyb_double_hoist: the same plaintext's limbs over the special primes [ksp][N] (NTT form; in an arena), or null
};

struct KeySet {
    u64 *sk = nullptr;    // [K][N]
    u64 *pk = nullptr;    // [2][K][N]
    u64 *relin = nullptr; // [K-1][2][K][N]
    std::map<u32, u64 *> galois;
};

class HEVM {
  public:
    std::unique_ptr<Context> ctx;
    std::unique_ptr<HostEncoder> encoder;
    KeySet keys;
    bool debug = false;
    void *ckks_handle = nullptr; // dc_context* handed out by hevm_context()
    RngKeys rng;          // ChaCha20 keys of this VM's randomness (chacha.hpp)
    u64 enc_counter = 0;  // encryptions made so far: the `object` of the next one
    u64 *d_epoch = nullptr; // run() counter in HBM, mixed into the encryption randomness (graph replays stay fresh)

    // The VM's HIP stream and the scratch of one in-flight composite op.  run() executes the batched plan (plan.hpp); with
    // option plan = 0 or setDebug(true) it is the reference's loop instead: one instruction at a time, in program order.
    struct Lane {
        hipStream_t stream = nullptr;
        Workspace ws;
        Plain boot_plain; // staging plaintext of opcode 10
    };
    std::vector<Lane> lanes; // one entry
    int cur = 0;
    hipStream_t S() const { return lanes[cur].stream; }
    const Workspace &W() const { return lanes[cur].ws; }
    void execute();
    void dispatch(const WireOp &op);

    // program
    std::vector<std::vector<double>> buffer; // constants of the .cst file
    // batched device encoder (encoder.hip): tables, and the arenas the plaintext registers of the loaded program live in
    EncTables enc_tables;
    std::vector<u64 *> plain_arenas;
    // option online_encode = 1: plaintexts are encoded at use from the resident constants instead of being kept pre-encoded
    bool online_encode = false;
    struct OnlineEncode {
        double *d_consts = nullptr;
        size_t const_bytes = 0;
        std::map<int, EncItem> items; // plaintext register -> what to encode
    } online;
    bool host_encoder = false; // option host_encoder = 1: encode on the host (HostEncoder), one plaintext at a time
    void ensure_enc_tables();
    void preprocess_device();
    void free_plains();
    WireHeader header{};
    WireConfigBody config{};
    std::vector<WireOp> ops;
    std::vector<uint64_t> arg_scale, arg_level, res_scale, res_level, res_dst;
    std::deque<hevm_ctxt> ciphers; // deque: growing it never invalidates references to registers
    std::vector<Plain> plains;

    // device CRT tables per level + staging plaintext of opcode 10
    struct CrtTables {
        u64 *inv, *mmod, *hmod, *hdig;
        double *mdbl;
    };
    std::map<int, CrtTables> crt_;
    const CrtTables &crt_tables(int ell);

    // ---- batched plan (plan.hpp): built on the first run() of a loaded program -----------------------------
    enum PopKind { P_ROT, P_MULCC, P_RESCALE, P_SUM, P_NEG, P_MULP, P_ADDP, P_COPY, P_BOOT, P_MODRAISE, P_ROTSUM };
    static constexpr int kPopKinds = 11;
    struct Val {
        int level = 0;
        double scale = 1.0;
        int root = -1;     // value whose buffer this one aliases (modswitch = a view with fewer limbs)
        int def_step = -1; // step producing it (-1: program input)
        int last_use = -1; // last step reading it
        int uses = 0;
        int def_pop = -1;
        u64 *buf = nullptr;
        bool external = false, pinned = false;
    };
    struct Pop {
        PopKind kind;
        int level = 0;
        int dst = -1;
        std::vector<int> srcs;
        std::vector<int> src_plain; // P_SUM: plain register multiplying srcs[k], or -1
        u32 elt = 0;
        const u64 *key = nullptr;
        int plain = -1, target_level = 0;
        int rs_add = -1, rs_mul = -1; // P_RESCALE: plain registers of a folded addcp / mulcp (operand = (srcs + add) * mul)
        bool rs_sum = false;          // P_RESCALE: srcs/src_plain are the terms of a folded n-ary sum
        int boot_drop = -1;           // P_BOOT: prime index of a folded rescale (operand read before the division), or -1
        double boot_src_scale = 0.0;  // P_BOOT with boot_drop: scale of the folded rescale's result
        bool dead = false;
        int wave = 0, step = -1;
        // lazy sums (option hyb_lazy_sum): a P_ROT remembers the instruction it came from and whether it is that instruction's LAST hop;
        // a P_ROTSUM (dst = sum of galois_{elts[k]}(srcs[k]) with one division by P, plan.hpp hyb_rotate_sum) lists its rotations
        int op = -1;
        bool direct = false;
        std::vector<u32> elts;
        std::vector<const u64 *> keys;
        std::vector<int> ops;
        std::vector<int> plains; // P_ROTSUM, option hyb_double_hoist: the plaintext register multiplying rotation k before it joins the sum, or -1
    };
    struct Step {
        PopKind kind;
        int level = 0, first = 0, count = 0; // range in the kind's device item table
        int target = 0;                     // P_BOOT: primes of the result
        int wave = 0, lane = 0;             // steps of one wave are independent: lane 1 runs on the auxiliary stream
        Handoff h;                          // link to a fused producer (h.in) / consumer (h.cont, h.out) step, plan.hpp
        int fused_consumer = -1;            // index of the step whose first phase this step's last kernel computes
        int gfirst = 0, gcount = 0;         // P_SUM: the step's items as groups that share sources (plan.hpp SumGroup), when it runs that way; P_ROTSUM: its groups (in d_ks)
        int unique = 0;                     // P_ROT, grouped-digit mode: distinct source ciphertexts among the items (shared decompositions)
        // what the step's launches read and write, by pool buffer (a value and its modswitch views share one): the edges of the explicitly
        // built graph (option plan_graph = 2, capture_plan_dag)
        std::vector<const u64 *> reads, writes;
    };
    struct Plan {
        bool ready = false;
        std::vector<Val> vals;
        std::vector<Pop> pops;
        std::vector<Step> steps;
        std::vector<int> final_val; // architectural register -> value at program end
        KsItem *d_ks = nullptr;
        MulItem *d_mul = nullptr;
        RsItem *d_rs = nullptr;
        EwItem *d_ew = nullptr;
        SumItem *d_sum = nullptr;
        SumSrc *d_sum_srcs = nullptr;
        SumGroup *d_sumg = nullptr;
        SumGroupSrc *d_sumg_srcs = nullptr;
        CtView *d_cont_other = nullptr;      // CONT_MUL links: the consumers' other operands
        // on-line encode: per wave, the plaintext registers first read in it, encoded into a window recycled at wave granularity
        struct EncGroup { int wave, level, first, count; u64 *out; };
        std::vector<EncGroup> enc_groups;
        EncItem *d_enc_items = nullptr;
        u64 *enc_arena = nullptr;
        double2 *enc_scratch = nullptr;
        int *d_enc_overflow = nullptr;
        size_t enc_arena_bytes = 0, enc_scratch_bytes = 0;
        std::vector<EwItem> h_ew;            // host copy of the elementwise item table (modraise launches its NTTs per item)
        std::vector<u64 *> handoff_bufs;     // first-phase buffers of fused consumer steps
        size_t n_fused = 0;
        // opcode 10: item table, the divide-and-round items of the zero-encryptions, the zero-encryption arena and scratch
        BootItem *d_boot = nullptr;
        RsItem *d_boot_rs = nullptr;
        size_t zenc_bytes = 0;
        u64 *zenc = nullptr, *boot_ue = nullptr, *boot_tmp = nullptr, *boot_pt[2] = { nullptr, nullptr }, *boot_ptx[2] = { nullptr, nullptr };
        struct BootChunk { int first, count, target; };
        std::vector<BootChunk> boot_chunks; // zero-encryption launches at the start of every run
        BatchWs ws[2]; // per lane
        std::vector<hipEvent_t> events; // fork/join pairs of the waves that use the auxiliary stream
        std::vector<u64 *> pool; // every pool buffer ever allocated (reused across plans)
        int64_t n_keyswitch = 0, n_ntt = 0;
        size_t launches = 0, max_live = 0;
        hipGraph_t graph = nullptr;
        hipGraphExec_t graph_exec = nullptr;
    } plan;
    bool plan_graph = true; // replay the plan's launch sequence as one HIP graph (option plan_graph = 0: issue it launch by launch)
    bool plan_dag = false;  // option plan_graph = 2: the graph is BUILT from the plan's own dependencies instead of captured from two streams
    void capture_plan();
    bool capture_plan_dag();
    void issue_step(const Step &st, hipStream_t q);
    void drop_plan_graph();
    void issue_plan(hipStream_t s);
    bool use_plan = true;
    bool test_zero_enc = false; // hevm_test_zero_encryption: encryptions of zero are (0, 0) -- INSECURE, parity tests of opcode 10 only
    int max_batch = 64; // items per heavy step (option max_batch): 64 measured best on the ResNet-20 program (16 ... 256 tried)
    std::vector<u64 *> home; // permanent buffer block of every architectural register (program inputs live here)
    // Throughput mode: `streams` independent ciphertext streams share the program, keys and plaintexts; every buffer is a
    // block of `streams` slices and encrypt/decrypt/getCtxt address the slice selected by hevm_select_stream().
    int streams = 1, sel = 0;
    std::vector<u64 *> reg_base; // base of the block currently holding each architectural register
    void set_streams(int n);
    void select_stream(int s);
    void build_plan();
    void run_plan();
    void boot_item(CtView src, int src_level, double src_scale, hevm_ctxt &dst, int target_level);
    void plan_zero_encrypt(int first, int B, int t, hipStream_t s);
    void plan_boot_step(int first, int B, int ell, int t, int lane, hipStream_t s, const Handoff &h);
    bool rot_compose = false; // option rot_compose: rotations without a direct key are the shortest sum of offsets that have one (HEVM::compose_rotation)
    std::vector<int> compose_rotation(int steps) const;
    mutable std::vector<int> rot_offsets;   // offsets with a key, in the search order; rebuilt when the key set changes
    mutable std::set<int> rot_offset_set;
    mutable size_t rot_offsets_epoch = (size_t)-1;
    int secret_weight = 0; // option secret_hw = h: key generation draws a ternary secret with exactly h non-zero coefficients (0: uniform ternary, SEAL's)
    bool chain_fusion = true; // option chain_fusion = 0: every step runs all of its own launches
    hipStream_t aux_stream = nullptr;
    // option hyb_double_hoist = 1 (with hyb_lazy_sum): rotations multiplied by a plaintext join the lazy sums; the plaintexts' special-prime limbs
    // are encoded when the plan first names them (the program's constants and encode items stay resident for that)
    bool double_hoist = false;
    double *dh_consts = nullptr;
    std::map<int, EncItem> dh_items; // plaintext register -> its encode item (preprocess_device)
    void ensure_special_limbs(const std::vector<int> &plain_regs);
    bool lazy_sums = false; // option hyb_lazy_sum = 1: sums of direct-key rotations share one division by P (plan_exec.hip section 2b)
    bool fold_rescale_into_boot = false; // option fold_rescale_boot = 1: do a rescale that only feeds an opcode 10 inside its re-encoder
    int plan_lanes = 2; // independent steps of a wave also use an auxiliary stream (pays off only inside the graph; option plan_lanes = 1: one stream)
    void bump_epoch(hipStream_t s);

    // statistics of the last run()
    int64_t op_counts[11] = { 0 };
    int64_t n_keyswitch = 0, n_ntt = 0;
    double t_bootstrap = 0.0; // host wall time spent inside opcode 10

    VmAllocs allocs;   // device memory held by this VM (hevm_destroy)
    HEVM() { g_vm_allocs = &allocs; }
    void destroy_device_state(); // streams, events, graph, every tracked allocation; the object is unusable afterwards
    size_t key_elems() const { return (size_t)ctx->key_digits() * 2 * ctx->K * ctx->N; }
    void init_context(int logN, int K, const u64 *primes, int dir_ksp = 0, int dir_alpha = 0);
    void generate_keys(const RngKeys &rng, bool secret, bool pub, bool eval);
    void gen_kswitch_key(u64 *key, const u64 *new_key, u64 key_id);
    void add_galois_key(u32 elt);
    void save_keys(const std::string &dir);
    void load_keys(const std::string &dir, bool need_secret, bool need_public, bool need_eval);
    sealio::ParmsId parms_id_at(int limbs) const; // SEAL parms_id of the chain truncated to `limbs` primes (limbs = K: key level)
    void save_ctxt(size_t reg, const std::string &path);
    void load_ctxt(size_t reg, const std::string &path);

    void load_constants(const void *data, size_t len);
    void load_program(const void *data, size_t len, bool header_only);
    void reset_res_dst();
    void preprocess();
    void encode_internal(Plain &dst, const double *src, size_t len, int level, int scale_bits);
    hevm_ctxt &reg(size_t i);
    void encrypt_plain(hevm_ctxt &dst, const Plain &pt);
    void encrypt(int64_t i, const double *dat, int len);
    void decrypt(int64_t i, double *out);
    void run();

    // opcode handlers (SEAL_HEVM.cpp:268-334)
    void op_rotate(int dst, int src, int offset);
    void op_negate(int dst, int src);
    void op_rescale(int dst, int src);
    void op_modswitch(int dst, int src, int down);
    void op_addcc(int dst, int lhs, int rhs);
    void op_addcp(int dst, int lhs, int rhs);
    void op_mulcc(int dst, int lhs, int rhs);
    void op_mulcp(int dst, int lhs, int rhs);
    void op_bootstrap(int dst, int src, int target_level);
    void op_conj(int dst, int src);
    void op_modraise(int dst, int src, int target);
    void op_setscale(int dst, int src, int const_idx);
    std::vector<u32> rotate_hops(int steps) const;

    CtView view(const hevm_ctxt &c) const { return CtView{ c.data, (long)c.poly_stride }; }
};

} // namespace dacapo
