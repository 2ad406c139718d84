// abi_common.h -- host-side helpers shared by the files that implement the C ABI
// (abi.hip: device entry points and small host helpers; host_api.hip: host-pointer entry
// points and the lz4hip_stream handle).  Host code only.
#pragma once
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace lz4hip {

extern thread_local std::string g_err;  // text behind lz4hip_last_error()

inline int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
inline int hip_fail(hipError_t e, const char *what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return LZ4HIP_ERR_HIP;
}
#define HIPCHK(call)                                      \
    do {                                                  \
        hipError_t e_ = (call);                           \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

inline bool valid_block_size(uint32_t bs) {
    return bs == LZ4HIP_BLOCK_64KB || bs == LZ4HIP_BLOCK_256KB || bs == LZ4HIP_BLOCK_1MB ||
           bs == LZ4HIP_BLOCK_4MB;
}
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr size_t kFastSlots = 8192;             // resident waves of the global-table Fast kernel
constexpr size_t kFastSlotBytes = 65536 * 2;    // one u16 table
constexpr size_t kFastCounterBytes = 256;       // tail of the Fast workspace: the block counter of the dynamic schedule
#ifndef LZ4HIP_FAST_GLOBAL_MIN
#define LZ4HIP_FAST_GLOBAL_MIN 513
#endif
constexpr size_t kFastGlobalMinBlocks = LZ4HIP_FAST_GLOBAL_MIN;    // below: the team kernel (table in LDS, one block per CU)
constexpr size_t kHcSlots = 4096;               // resident workgroups of the HC kernel
constexpr uint32_t kLegacyBlock = 8u << 20;     // blocks.go:11, frame.go:49
constexpr size_t kMaxSingleBlock = 1u << 30;    // int32 positions inside a block

// A Fast workspace is `slots` hash tables followed, when the caller sized it with
// lz4hip_fast_workspace_bytes, by the block counter of the dynamic schedule.  A workspace
// without room for the counter still works: blocks are then dealt round-robin.
inline uint32_t *fast_ws_split(void *ws, size_t bytes, size_t &slots) {
    slots = ws ? bytes / kFastSlotBytes : 0;
    if (slots > kFastSlots) slots = kFastSlots;
    if (!slots || bytes - slots * kFastSlotBytes < sizeof(uint32_t)) return nullptr;
    return reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(ws) + slots * kFastSlotBytes);
}

// The host-pointer entry points run on the calling thread's own stream
// (hipStreamPerThread), never on the legacy null stream: calls made from several OS
// threads (cgo runs each goroutine's call on its own thread) then overlap on the GPU
// instead of queueing behind each other.  hcopy = a blocking copy on that stream.
const hipStream_t kHostStream = hipStreamPerThread;
inline hipError_t hcopy(void *dst, const void *src, size_t n, hipMemcpyKind kind) {
    hipError_t e = hipMemcpyAsync(dst, src, n, kind, kHostStream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(kHostStream);
}

// RAII device buffer for the host-pointer entry points: stream-ordered allocation on the
// calling thread's stream.  hipMalloc/hipFree would do, but hipFree waits for EVERY stream
// of the device, which serialises concurrent callers behind each other's kernels.
// Every use of the buffer on another stream is synchronised before the owner returns.
// After lz4hip_init() freed memory stays in the device's pool, so a call's buffers cost
// no driver allocation (the reference keeps its buffers in sync.Pools the same way).
struct DevBuf {
    void *p = nullptr;
    hipError_t alloc(size_t n) { return hipMallocAsync(&p, n ? n : 16, kHostStream); }
    ~DevBuf() {
        if (p) (void)hipFreeAsync(p, kHostStream);
    }
    template <class T> T *as() { return static_cast<T *>(p); }
};

// A second, non-blocking stream owned by one call.  The destructor waits for it, so
// buffers declared BEFORE it outlive everything queued on it -- on every exit path.
struct SideStream {
    hipStream_t s = nullptr;
    hipError_t create() { return hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
    ~SideStream() {
        if (s) {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    }
};

// Large copies between the caller's (pageable) memory and the device go through two pinned
// buffers that a few host threads fill or drain while the other one is on the bus: on the
// MI355X box hipMemcpy from pageable memory runs at 29 GB/s up and 17.5 GB/s down, the staged
// copy at 50-55 GB/s either way (tools/micro/xfer.hip, profiles/r2_host_xfer.txt).  A Stager
// belongs to one call at a time (Stager::Lease takes one from a per-device free list) and
// keeps its 2 x 32 MiB for the next call.
class Stager {
public:
    static constexpr size_t kPiece = 32u << 20, kDirect = 4u << 20;
    static constexpr int kThreads = 4;

    // Queues h[0,n) -> d on st.  Returns once h has been read; the last pieces may still be
    // on the bus (work queued on st afterwards is ordered behind them).
    hipError_t up(void *d, const void *h, size_t n, hipStream_t st) {
        if (n <= kDirect) return n ? hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st) : hipSuccess;
        hipError_t e = ready();
        int k = 0;
        for (size_t a = 0; a < n && e == hipSuccess; a += kPiece, k ^= 1) {
            const size_t m = std::min(kPiece, n - a);
            if ((e = hipEventSynchronize(ev_[k])) != hipSuccess) break;  // its last copy has left
            spread(pin_[k], static_cast<const uint8_t *>(h) + a, m);
            if ((e = hipMemcpyAsync(static_cast<uint8_t *>(d) + a, pin_[k], m, hipMemcpyHostToDevice, st)) != hipSuccess) break;
            e = hipEventRecord(ev_[k], st);
        }
        return e;
    }
    // d[0,n) -> h on st, complete on return; arrived(upto) is called as h[0,upto) becomes final.
    template <class F> hipError_t down(void *h, const void *d, size_t n, hipStream_t st, F &&arrived) {
        hipError_t e = hipSuccess;
        if (n <= kDirect) {
            if (n) e = hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e == hipSuccess && n) arrived(n);
            return e;
        }
        if ((e = ready()) != hipSuccess) return e;
        size_t at[2] = {0, 0}, len[2] = {0, 0};
        int k = 0;
        for (size_t a = 0; a < n + kPiece && e == hipSuccess; a += kPiece, k ^= 1) {
            if (len[k]) {  // the piece fetched into this buffer two turns ago
                if ((e = hipEventSynchronize(ev_[k])) != hipSuccess) break;
                spread(static_cast<uint8_t *>(h) + at[k], pin_[k], len[k]);
                arrived(at[k] + len[k]);
                len[k] = 0;
            }
            if (a < n) {
                const size_t m = std::min(kPiece, n - a);
                // (an upload of an earlier call may still be reading this buffer)
                if ((e = hipEventSynchronize(ev_[k])) != hipSuccess) break;
                if ((e = hipMemcpyAsync(pin_[k], static_cast<const uint8_t *>(d) + a, m, hipMemcpyDeviceToHost, st)) != hipSuccess) break;
                e = hipEventRecord(ev_[k], st);
                at[k] = a, len[k] = m;
            }
        }
        if (e == hipSuccess && len[k]) {  // an even count of pieces leaves the last one here
            if ((e = hipEventSynchronize(ev_[k])) == hipSuccess) {
                spread(static_cast<uint8_t *>(h) + at[k], pin_[k], len[k]);
                arrived(at[k] + len[k]);
            }
        }
        if (e != hipSuccess) (void)hipStreamSynchronize(st);  // nothing may land in pin_ after the lease ends
        return e;
    }
    hipError_t down(void *h, const void *d, size_t n, hipStream_t st) {
        return down(h, d, n, st, [](size_t) {});
    }
    // d[0,n) handed to piece(ptr, len) in order, in the pinned buffers themselves (no copy to
    // pageable memory): one piece is on the bus while the caller works on the other.
    template <class F> hipError_t drain(const void *d, size_t n, hipStream_t st, F &&piece) {
        if (!n) return hipSuccess;
        hipError_t e = ready();
        if (e != hipSuccess) return e;
        const uint8_t *src = static_cast<const uint8_t *>(d);
        const size_t pieces = (n + kPiece - 1) / kPiece;
        auto fetch = [&](size_t k) -> hipError_t {
            const size_t at = k * kPiece, len = std::min(kPiece, n - at);
            hipError_t f = hipEventSynchronize(ev_[k & 1]);
            if (f == hipSuccess) f = hipMemcpyAsync(pin_[k & 1], src + at, len, hipMemcpyDeviceToHost, st);
            return f != hipSuccess ? f : hipEventRecord(ev_[k & 1], st);
        };
        e = fetch(0);
        for (size_t k = 0; k < pieces && e == hipSuccess; k++) {
            // piece k+1 goes into the buffer that piece k-1 was consumed from
            if (k + 1 < pieces && (e = fetch(k + 1)) != hipSuccess) break;
            if ((e = hipEventSynchronize(ev_[k & 1])) != hipSuccess) break;
            piece(pin_[k & 1], std::min(kPiece, n - k * kPiece));
        }
        if (e != hipSuccess) (void)hipStreamSynchronize(st);
        return e;
    }

    // One Stager for the duration of a call, from the free list of the current device.
    class Lease {
    public:
        Lease() {
            if (hipGetDevice(&dev_) != hipSuccess) dev_ = 0;
            std::lock_guard<std::mutex> g(mu());
            auto &v = idle()[dev_ & 63];
            if (!v.empty()) {
                s_ = v.back();
                v.pop_back();
            }
            if (!s_) s_ = new Stager();
        }
        ~Lease() {
            // up() returns once the host bytes have been read; its last pieces may still be on
            // the bus out of pin_[].  The next lessee must not find them in use.
            s_->settle();
            std::lock_guard<std::mutex> g(mu());
            idle()[dev_ & 63].push_back(s_);
        }
        Stager *operator->() { return s_; }

    private:
        Stager *s_ = nullptr;
        int dev_ = 0;
    };
    // lz4hip_shutdown(): waits for and releases every idle stager of every device.
    static void release_all() {
        std::lock_guard<std::mutex> g(mu());
        for (int i = 0; i < 64; i++) {
            auto &v = idle()[i];
            for (Stager *s : v) delete s;
            v.clear();
        }
    }
    ~Stager() {
        for (int i = 0; i < 2; i++) {
            if (ev_[i]) {
                (void)hipEventSynchronize(ev_[i]);
                (void)hipEventDestroy(ev_[i]);
            }
            if (pin_[i]) (void)hipHostFree(pin_[i]);
        }
    }

    void settle() {
        for (int i = 0; i < 2; i++)
            if (ev_[i]) (void)hipEventSynchronize(ev_[i]);
    }

private:
    static std::mutex &mu() {
        static std::mutex m;
        return m;
    }
    static std::vector<Stager *> *idle() {
        static std::vector<Stager *> v[64];
        return v;
    }
    hipError_t ready() {
        for (int i = 0; i < 2; i++) {
            hipError_t e;
            if (!pin_[i] && (e = hipHostMalloc((void **)&pin_[i], kPiece, hipHostMallocPortable)) != hipSuccess) return e;
            if (!ev_[i] && (e = hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming)) != hipSuccess) return e;
        }
        return hipSuccess;
    }
    static void spread(uint8_t *d, const uint8_t *s, size_t n) {  // memcpy on kThreads threads
        const size_t per = ((n / kThreads) + 4095) & ~(size_t)4095;
        std::thread t[kThreads];
        int used = 0;
        for (int i = 1; i < kThreads && (size_t)i * per < n; i++, used++) {
            const size_t a = (size_t)i * per, m = std::min(per, n - a);
            t[i] = std::thread([=] { memcpy(d + a, s + a, m); });
        }
        memcpy(d, s, std::min(per, n));
        for (int i = 1; i <= used; i++) t[i].join();
    }
    uint8_t *pin_[2] = {nullptr, nullptr};
    hipEvent_t ev_[2] = {nullptr, nullptr};
};

// ---------------------------------------------------------------------------------
// XXHZero on the host: internal/xxh32/xxh32zero.go:22-139 restated, Sum32's
// `uint32(totalLen) >= 16` included (:113-118, SURVEY trap T6).  Product code: the
// host-pointer frame entry points and the lz4hip_stream handle compute the CONTENT
// checksum here -- the bytes are in host memory on those paths anyway, and one host core
// runs this serial chain four times faster than a GPU lane chain does.
// ---------------------------------------------------------------------------------
struct HostXXH {
    uint32_t v[4];
    uint64_t total;
    uint8_t buf[16];
    int used;

    static uint32_t rol(uint32_t u, int k) { return (u << k) | (u >> (32 - k)); }
    static uint32_t rd(const uint8_t *p) {
        uint32_t w;
        memcpy(&w, p, 4);
        return w;
    }
    HostXXH() { reset(); }
    void reset() {  // :37-44
        v[0] = 606290984u;
        v[1] = 2246822519u;
        v[2] = 0u;
        v[3] = 1640531535u;
        total = 0;
        used = 0;
    }
    static void stripes(uint32_t *v, const uint8_t *p, size_t n16) {  // updateGo, :90-109
        uint32_t a = v[0], b = v[1], c = v[2], d = v[3];
        for (size_t i = 0; i < n16; i++, p += 16) {
            a = rol(a + rd(p) * 2246822519u, 13) * 2654435761u;
            b = rol(b + rd(p + 4) * 2246822519u, 13) * 2654435761u;
            c = rol(c + rd(p + 8) * 2246822519u, 13) * 2654435761u;
            d = rol(d + rd(p + 12) * 2246822519u, 13) * 2654435761u;
        }
        v[0] = a, v[1] = b, v[2] = c, v[3] = d;
    }
    void write(const uint8_t *in, size_t n) {  // :58-86
        if (total == 0) reset();
        total += n;
        const size_t room = 16 - (size_t)used;
        if (n < room) {
            if (n) memcpy(buf + used, in, n);
            used += (int)n;
            return;
        }
        if (used) {
            memcpy(buf + used, in, room);
            stripes(v, buf, 1);
            in += room;
            n -= room;
        }
        stripes(v, in, n / 16);
        used = (int)(n % 16);
        if (used) memcpy(buf, in + n - used, (size_t)used);
    }
    uint32_t sum32() const {  // :112-139
        uint32_t h = (uint32_t)total;
        if (h >= 16)
            h += rol(v[0], 1) + rol(v[1], 7) + rol(v[2], 12) + rol(v[3], 18);
        else
            h += 374761393u;
        int p = 0;
        for (; p + 4 <= used; p += 4) h = rol(h + rd(buf + p) * 3266489917u, 17) * 668265263u;
        for (; p < used; p++) h = rol(h + buf[p] * 374761393u, 11) * 2654435761u;
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        h *= 3266489917u;
        h ^= h >> 16;
        return h;
    }
};

// The content checksum of one call, on a helper thread that follows the producer's
// progress mark: lz4stream/block.go:257-259 (Write folds every block's src in stream
// order) and :336-338 / :174-176 (the reader's collector does the same with decoded
// blocks).  The producer publishes "bytes [0, upto) of base are final"; finish() waits
// for the hash of everything published.  The state it updates outlives the call
// (lz4hip_stream) or is local to it (frame entry points).
class HashFollower {
public:
    HashFollower(HostXXH *st, const uint8_t *base) : st_(st), base_(base) {
        th_ = std::thread([this] { run(); });
    }
    ~HashFollower() { finish(); }
    void publish(size_t upto) {
        {
            std::lock_guard<std::mutex> g(mu_);
            avail_ = upto;
        }
        cv_.notify_one();
    }
    void finish() {
        if (!th_.joinable()) return;
        {
            std::lock_guard<std::mutex> g(mu_);
            done_ = true;
        }
        cv_.notify_one();
        th_.join();
    }

private:
    void run() {
        size_t at = 0;
        for (;;) {
            size_t upto;
            bool done;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return avail_ > at || done_; });
                upto = avail_;
                done = done_;
            }
            if (upto > at) {
                st_->write(base_ + at, upto - at);
                at = upto;
            } else if (done) {
                return;
            }
        }
    }
    HostXXH *st_;
    const uint8_t *base_;
    std::mutex mu_;
    std::condition_variable cv_;
    size_t avail_ = 0;
    bool done_ = false;
    std::thread th_;
};

}  // namespace lz4hip
