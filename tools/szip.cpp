// szip -- gzip-like front end for the Snappy frame format on an MI355X.
//
// The counterpart of the reference's szip (szip/main.rs:1-260: same flags,
// same file naming, same stream bytes) over the host-buffer entry points of
// libsnapmi.so.  The reference pushes one 64 KiB chunk at a time through
// snap::write::FrameEncoder / snap::read::FrameDecoder; here the stream is
// cut into slabs of whole chunks (64 MiB) that flow through a pipeline
//
//     reader  ->  N workers, one snapmi_ctx (HIP stream) each  ->  writer
//
// so that the H2D copy of slab i+1, the kernels of slab i and the D2H copy /
// file write of slab i-1 overlap.  Slabs leave in order; a decoding error
// stops the stream behind the bytes of the chunks in front of it, like the
// reference's reader (src/read.rs:111-118).
//
//   szip [-d] [-f] [-k] [-r] [-j N] [-v] [FILE...]     (stdin -> stdout without FILE)
#include <sys/stat.h>
#include <unistd.h>
#include <utime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "snapmi.h"

namespace {

constexpr size_t kChunk = 65536;          // reference src/lib.rs:97
constexpr size_t kSlab = 64u << 20;       // input bytes per device call

struct Options {
    bool decompress = false, force = false, keep = false, raw = false,
         verbose = false;
    int workers = 2;
};

struct Buf { // page-locked when possible
    uint8_t *p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    void reserve(size_t n)
    {
        if (n <= cap)
            return;
        release();
        p = (uint8_t *)snapmi_host_alloc(n);
        pinned = p != nullptr;
        if (!p)
            p = (uint8_t *)malloc(n);
        cap = n;
    }
    void release()
    {
        if (p)
            pinned ? snapmi_host_free(p) : free(p);
        p = nullptr;
        cap = 0;
    }
    ~Buf() { release(); }
};

struct Job {
    uint64_t seq = 0;
    Buf in, out;
    size_t in_len = 0, out_len = 0;
    uint32_t flags = 0;       // decode: CONTINUATION / FINAL
    // compress from a regular file: the worker reads [file_off, +in_len)
    // itself (page cache -> pinned buffer is a memcpy; N workers, N copies)
    int fd = -1;
    uint64_t file_off = 0;
    uint8_t stale[10] = {0};  // decode: reader state in front of this slab
    size_t n_chunks = 0;      // decode: data chunks in the slab
    int rc = 0;               // result
    snapmi_error err{};
    std::string msg;
};

template <class T> class Queue {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<T> q;
    bool closed = false;

  public:
    void push(T v)
    {
        {
            std::lock_guard<std::mutex> l(mu);
            q.push_back(std::move(v));
        }
        cv.notify_one();
    }
    void close()
    {
        {
            std::lock_guard<std::mutex> l(mu);
            closed = true;
        }
        cv.notify_all();
    }
    bool pop(T &v)
    {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return !q.empty() || closed; });
        if (q.empty())
            return false;
        v = std::move(q.front());
        q.pop_front();
        return true;
    }
};

size_t read_full(FILE *f, uint8_t *p, size_t n)
{
    size_t got = 0;
    while (got < n) {
        size_t k = fread(p + got, 1, n - got, f);
        if (k == 0)
            break;
        got += k;
    }
    return got;
}

// One stream through the pipeline.  Returns 0, or 1 after printing why.
int run_stream(const Options &opt, FILE *src, FILE *dst, const char *name)
{
    if (opt.raw) {
        // the raw format has no framing: the whole input is one call
        // (reference szip/main.rs:216-222,238-244)
        std::vector<uint8_t> in;
        uint8_t tmp[1 << 16];
        size_t k;
        while ((k = fread(tmp, 1, sizeof tmp, src)) > 0)
            in.insert(in.end(), tmp, tmp + k);
        snapmi_ctx *ctx = nullptr;
        if (snapmi_ctx_create(0, nullptr, &ctx) != SNAPMI_OK) {
            fprintf(stderr, "szip: %s: no usable GPU\n", name);
            return 1;
        }
        snapmi_error e{};
        size_t cap = 0, n = 0;
        int rc;
        if (opt.decompress) {
            rc = snapmi_decompress_len(in.data(), in.size(), &cap, &e);
            std::vector<uint8_t> out(cap ? cap : 1);
            if (rc == SNAPMI_OK)
                rc = snapmi_raw_decompress(ctx, in.data(), in.size(),
                                           out.data(), cap, &n, &e);
            if (rc == SNAPMI_OK)
                fwrite(out.data(), 1, n, dst);
        } else {
            cap = snapmi_max_compress_len(in.size());
            std::vector<uint8_t> out(cap ? cap : 1);
            rc = cap || in.empty()
                     ? snapmi_raw_compress(ctx, in.data(), in.size(),
                                           out.data(), cap, &n, &e)
                     : (int)SNAPMI_TOO_BIG;
            if (rc == SNAPMI_OK)
                fwrite(out.data(), 1, n, dst);
        }
        if (rc != SNAPMI_OK) { // the reference's text (src/error.rs:249-335)
            char text[256];
            e.kind = rc;
            snapmi_error_string(&e, text, sizeof text);
            fprintf(stderr, "szip: %s: %s\n", name, text);
        }
        snapmi_ctx_destroy(ctx);
        return rc != SNAPMI_OK;
    }

    Queue<Job *> todo, done, pool;
    const int n_jobs = opt.workers + 2;
    std::vector<Job> jobs(n_jobs);
    for (auto &j : jobs)
        pool.push(&j);
    std::atomic<bool> failed{false};

    std::vector<std::thread> workers;
    for (int w = 0; w < opt.workers; w++)
        workers.emplace_back([&] {
            snapmi_ctx *ctx = nullptr;
            const int crc = snapmi_ctx_create(0, nullptr, &ctx);
            Job *j;
            while (todo.pop(j)) {
                if (crc != SNAPMI_OK) {
                    j->rc = SNAPMI_E_DEVICE;
                    j->msg = "no usable GPU";
                } else if (!opt.decompress) {
                    if (j->fd >= 0) {
                        j->in.reserve(kSlab);
                        size_t got = 0;
                        while (got < j->in_len) {
                            const ssize_t k = pread(j->fd, j->in.p + got,
                                                    j->in_len - got,
                                                    (off_t)(j->file_off + got));
                            if (k <= 0)
                                break;
                            got += (size_t)k;
                        }
                        if (got != j->in_len) {
                            j->rc = SNAPMI_E_ARGUMENT;
                            j->msg = "short read";
                            done.push(j);
                            continue;
                        }
                    }
                    std::vector<uint32_t> lens((j->in_len + kChunk - 1) /
                                               kChunk, (uint32_t)kChunk);
                    if (j->in_len % kChunk)
                        lens.back() = (uint32_t)(j->in_len % kChunk);
                    j->out.reserve(snapmi_frame_encode_bound(j->in_len,
                                                             lens.size()));
                    j->rc = snapmi_frame_encode_host(
                        ctx, j->in.p, lens.data(), lens.size(), j->flags,
                        j->out.p, j->out.cap, &j->out_len);
                    if (j->rc)
                        j->msg = snapmi_last_error(ctx);
                } else {
                    j->out.reserve(j->n_chunks * kChunk + 64);
                    size_t used = 0;
                    j->rc = snapmi_frame_decode_host(
                        ctx, j->in.p, j->in_len, j->flags, j->stale, j->out.p,
                        j->out.cap, &j->out_len, &used, &j->err);
                    if (j->rc >= 100)
                        j->msg = snapmi_last_error(ctx);
                    else if (j->rc == SNAPMI_OK && used != j->in_len) {
                        j->rc = SNAPMI_E_DEVICE;
                        j->msg = "slab not consumed";
                    }
                }
                done.push(j);
            }
            if (ctx)
                snapmi_ctx_destroy(ctx);
        });

    std::thread writer([&] {
        std::map<uint64_t, Job *> held;
        uint64_t next = 0;
        Job *j;
        while (done.pop(j)) {
            held[j->seq] = j;
            while (!held.empty() && held.begin()->first == next) {
                Job *k = held.begin()->second;
                held.erase(held.begin());
                next++;
                if (!failed) {
                    // the bytes in front of an error are valid output
                    if (k->out_len &&
                        fwrite(k->out.p, 1, k->out_len, dst) != k->out_len) {
                        fprintf(stderr, "szip: %s: write failed\n", name);
                        failed = true;
                    }
                    if (k->rc != SNAPMI_OK) {
                        if (k->rc >= 100)
                            fprintf(stderr, "szip: %s: %s\n", name,
                                    k->msg.c_str());
                        else {
                            char text[256];
                            snapmi_error e = k->err;
                            e.kind = k->rc;
                            snapmi_error_string(&e, text, sizeof text);
                            fprintf(stderr, "szip: %s: %s\n", name, text);
                        }
                        failed = true;
                    }
                }
                pool.push(k);
            }
        }
    });

    // the reader (this thread): slabs of whole chunks
    uint64_t seq = 0;
    bool first = true;
    struct stat sst;
    const int sfd = fileno(src);
    if (!opt.decompress && fstat(sfd, &sst) == 0 && S_ISREG(sst.st_mode) &&
        ftell(src) == 0) {
        for (uint64_t off = 0; off < (uint64_t)sst.st_size && !failed;
             off += kSlab) {
            Job *j = nullptr;
            if (!pool.pop(j))
                break;
            j->fd = sfd;
            j->file_off = off;
            j->in_len = (size_t)((uint64_t)sst.st_size - off < kSlab
                                     ? (uint64_t)sst.st_size - off
                                     : kSlab);
            j->seq = seq++;
            j->rc = 0;
            j->out_len = 0;
            j->flags = off == 0 ? 0 : SNAPMI_FRAME_NO_IDENT;
            todo.push(j);
        }
    } else if (!opt.decompress) {
        for (;;) {
            Job *j = nullptr;
            if (!pool.pop(j))
                break;
            j->fd = -1;
            j->in.reserve(kSlab);
            j->in_len = read_full(src, j->in.p, kSlab);
            if (j->in_len == 0) {
                pool.push(j);
                break;
            }
            j->seq = seq++;
            j->rc = 0;
            j->out_len = 0;
            j->flags = first ? 0 : SNAPMI_FRAME_NO_IDENT;
            first = false;
            todo.push(j);
            if (j->in_len < kSlab || failed)
                break;
        }
    } else {
        std::vector<uint8_t> carry;
        uint8_t stale[10] = {0};
        bool eof = false;
        while (!eof && !failed) {
            Job *j = nullptr;
            if (!pool.pop(j))
                break;
            size_t want = kSlab + carry.size();
            for (;;) { // until the slab holds at least one whole chunk
                j->in.reserve(want);
                memcpy(j->in.p, carry.data(), carry.size());
                const size_t got = read_full(src, j->in.p + carry.size(),
                                             want - carry.size());
                j->in_len = carry.size() + got;
                eof = got < want - carry.size();
                if (j->in_len == 0)
                    break;
                uint64_t n = 0, used = 0;
                uint8_t st[10];
                memcpy(st, stale, 10);
                const int status = snapmi_frame_scan_host(
                    j->in.p, j->in_len, first ? 0 : SNAPMI_FRAME_CONTINUATION,
                    st, nullptr, 0, &n, &used);
                memcpy(j->stale, stale, 10);
                j->flags = first ? 0 : SNAPMI_FRAME_CONTINUATION;
                j->n_chunks = n;
                if (status == 0 || (status == 2 && !eof && used > 0)) {
                    // whole chunks [0, used); the tail waits for more input
                    carry.assign(j->in.p + used, j->in.p + j->in_len);
                    j->in_len = used;
                    memcpy(stale, st, 10);
                    break;
                }
                if (status == 2 && !eof) { // not one whole chunk: read more
                    carry.assign(j->in.p, j->in.p + j->in_len);
                    want = j->in_len + kSlab;
                    continue;
                }
                // a rejected or cut-off chunk follows: the device names the
                // error behind the output of the chunks in front of it
                j->flags |= SNAPMI_FRAME_FINAL;
                j->n_chunks = n;
                carry.clear();
                eof = true;
                break;
            }
            if (j->in_len == 0) {
                pool.push(j);
                break;
            }
            j->seq = seq++;
            j->rc = 0;
            j->out_len = 0;
            first = false;
            todo.push(j);
        }
    }
    todo.close();
    for (auto &t : workers)
        t.join();
    done.close();
    writer.join();
    return failed ? 1 : 0;
}

int do_file(const Options &opt, const std::string &path)
{
    struct stat st;
    if (stat(path.c_str(), &st) != 0) {
        perror(path.c_str());
        return 1;
    }
    if (S_ISDIR(st.st_mode)) {
        fprintf(stderr, "szip: %s: is a directory\n", path.c_str());
        return 1;
    }
    // reference new_path, szip/main.rs:183-205
    std::string out;
    const bool has_sz = path.size() > 3 &&
                        path.compare(path.size() - 3, 3, ".sz") == 0;
    if (opt.decompress) {
        if (!has_sz) {
            fprintf(stderr, "szip: %s: skipping uncompressed file\n",
                    path.c_str());
            return 1;
        }
        out = path.substr(0, path.size() - 3);
    } else {
        if (has_sz) {
            fprintf(stderr, "szip: %s: skipping compressed file\n",
                    path.c_str());
            return 1;
        }
        out = path + ".sz";
    }
    if (!opt.force && access(out.c_str(), F_OK) == 0) {
        fprintf(stderr, "szip: skipping, file already exists: %s\n",
                out.c_str());
        return 1;
    }
    FILE *src = fopen(path.c_str(), "rb");
    FILE *dst = src ? fopen(out.c_str(), "wb") : nullptr;
    if (!src || !dst) {
        perror(src ? out.c_str() : path.c_str());
        if (src)
            fclose(src);
        return 1;
    }
    setvbuf(dst, nullptr, _IOFBF, 1 << 20);
    const auto t0 = std::chrono::steady_clock::now();
    int rc = run_stream(opt, src, dst, path.c_str());
    fflush(dst);
    fclose(src);
    if (fclose(dst) != 0)
        rc = 1;
    if (opt.verbose) {
        const double s = std::chrono::duration<double>(
                             std::chrono::steady_clock::now() - t0).count();
        struct stat so;
        stat(out.c_str(), &so);
        const double u = (double)(opt.decompress ? so.st_size : st.st_size);
        fprintf(stderr, "szip: %s: %.3f s, %.2f GiB/s (uncompressed bytes)\n",
                path.c_str(), s, u / s / (1 << 30));
    }
    if (rc != 0) {
        unlink(out.c_str());
        return rc;
    }
    struct utimbuf tb = {st.st_atime, st.st_mtime}; // szip/main.rs:176-178
    utime(out.c_str(), &tb);
    if (!opt.keep)
        unlink(path.c_str());
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    Options opt;
    std::vector<std::string> paths;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-d" || a == "--decompress")
            opt.decompress = true;
        else if (a == "-f" || a == "--force")
            opt.force = true;
        else if (a == "-k" || a == "--keep")
            opt.keep = true;
        else if (a == "-r" || a == "--raw")
            opt.raw = true;
        else if (a == "-v" || a == "--verbose")
            opt.verbose = true;
        else if (a == "-j" && i + 1 < argc)
            opt.workers = atoi(argv[++i]) > 0 ? atoi(argv[i]) : 1;
        else if (a == "-h" || a == "--help") {
            puts("szip [-d] [-f] [-k] [-r] [-j workers] [-v] [FILE...]\n"
                 "  compresses FILE to FILE.sz (Snappy frame format) on the "
                 "GPU and removes FILE;\n  -d decompresses, -k keeps the "
                 "input, -f overwrites, -r raw format,\n  no FILE: stdin to "
                 "stdout.");
            return 0;
        } else if (!a.empty() && a[0] == '-' && a != "-") {
            fprintf(stderr, "szip: unknown option %s\n", a.c_str());
            return 2;
        } else
            paths.push_back(a);
    }
    if (paths.empty())
        return run_stream(opt, stdin, stdout, "<stdin>");
    int rc = 0;
    for (const auto &p : paths)
        rc |= do_file(opt, p);
    return rc;
}
