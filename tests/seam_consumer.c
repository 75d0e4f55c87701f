/* TEST INFRASTRUCTURE: a consumer of the reference's native seam, written
 * against snappy-c.h and linked with -lsnappy - the four functions the
 * reference's snappy-cpp crate binds (snappy-cpp/src/lib.rs:66-88) plus
 * snappy_validate_compressed_buffer - exactly as a program that uses Google's
 * libsnappy would be.  Which library answers is decided at link / load time
 * alone: tests/test_gpu_seam.py builds it once against a directory whose
 * libsnappy.so is a symlink to libsnapmi.so (no source change: the
 * snappy-cpp/build.rs:2 situation) and once against the real libsnappy 1.1.8.
 *
 *   seam_consumer check <dir>             every <name>.in of <dir>: compress
 *        must give the bytes of <name>.snappy, uncompress the input back;
 *        the error statuses of short buffers and broken streams
 *   seam_consumer bench <dir> <threads> <ms>   one call per file, the list of
 *        bench/src/bench.rs:83-153, <threads> callers at once for about <ms>
 *        milliseconds per file and direction: MB/s of uncompressed bytes
 */
#include <dirent.h>
#include <pthread.h>
#include <snappy-c.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + t.tv_nsec * 1e-9;
}

static char *slurp(const char *path, size_t *n)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *p = malloc(sz ? sz : 1);
    if (fread(p, 1, sz, f) != (size_t)sz) {
        fclose(f);
        free(p);
        return NULL;
    }
    fclose(f);
    *n = (size_t)sz;
    return p;
}

struct item {
    char name[256];
    char *in, *want;
    size_t n_in, n_want;
};
static struct item items[64];
static int n_items;

static int by_name(const void *a, const void *b)
{
    return strcmp(((const struct item *)a)->name,
                  ((const struct item *)b)->name);
}

static int load(const char *dir)
{
    DIR *d = opendir(dir);
    if (!d)
        return -1;
    struct dirent *e;
    while ((e = readdir(d)) && n_items < 64) {
        size_t l = strlen(e->d_name);
        if (l < 4 || strcmp(e->d_name + l - 3, ".in"))
            continue;
        struct item *it = &items[n_items];
        snprintf(it->name, sizeof it->name, "%.*s", (int)(l - 3), e->d_name);
        char path[1024];
        snprintf(path, sizeof path, "%s/%s.in", dir, it->name);
        it->in = slurp(path, &it->n_in);
        snprintf(path, sizeof path, "%s/%s.snappy", dir, it->name);
        it->want = slurp(path, &it->n_want);
        if (!it->in || !it->want)
            return -1;
        n_items++;
    }
    closedir(d);
    qsort(items, n_items, sizeof items[0], by_name);
    return n_items ? 0 : -1;
}

#define FAIL(...)                                                             \
    do {                                                                      \
        fprintf(stderr, "FAIL: " __VA_ARGS__);                                \
        fprintf(stderr, "\n");                                                \
        return 1;                                                             \
    } while (0)

static int check(void)
{
    for (int i = 0; i < n_items; i++) {
        struct item *it = &items[i];
        size_t cap = snappy_max_compressed_length(it->n_in), n = cap;
        if (cap != 32 + it->n_in + it->n_in / 6)
            FAIL("%s: max_compressed_length %zu", it->name, cap);
        char *c = malloc(cap + 64), *u = malloc(it->n_in + 64);
        memset(c, 0x5A, cap + 64);
        if (snappy_compress(it->in, it->n_in, c, &n) != SNAPPY_OK)
            FAIL("%s: snappy_compress", it->name);
        if (n != it->n_want || memcmp(c, it->want, n))
            FAIL("%s: compressed bytes differ (%zu vs %zu)", it->name, n,
                 it->n_want);
        for (size_t k = cap; k < cap + 64; k++)
            if ((unsigned char)c[k] != 0x5A)
                FAIL("%s: wrote behind the output buffer", it->name);
        size_t ul = 0;
        if (snappy_uncompressed_length(c, n, &ul) != SNAPPY_OK ||
            ul != it->n_in)
            FAIL("%s: uncompressed_length", it->name);
        if (snappy_validate_compressed_buffer(c, n) != SNAPPY_OK)
            FAIL("%s: validate", it->name);
        memset(u, 0x5A, it->n_in + 64);
        size_t un = it->n_in;
        if (snappy_uncompress(c, n, u, &un) != SNAPPY_OK || un != it->n_in ||
            memcmp(u, it->in, un))
            FAIL("%s: round trip", it->name);
        for (size_t k = it->n_in; k < it->n_in + 64; k++)
            if ((unsigned char)u[k] != 0x5A)
                FAIL("%s: wrote behind the decode buffer", it->name);
        /* statuses: short buffers, broken streams (snappy-c.h:49-69) */
        if (it->n_in > 8) {
            size_t small = cap - 1;
            if (snappy_compress(it->in, it->n_in, c, &small) !=
                SNAPPY_BUFFER_TOO_SMALL)
                FAIL("%s: compress into cap - 1", it->name);
            n = cap;
            snappy_compress(it->in, it->n_in, c, &n);
            un = it->n_in - 1;
            if (snappy_uncompress(c, n, u, &un) != SNAPPY_BUFFER_TOO_SMALL)
                FAIL("%s: uncompress into n - 1", it->name);
            un = it->n_in;
            if (snappy_uncompress(c, n / 2, u, &un) != SNAPPY_INVALID_INPUT)
                FAIL("%s: truncated stream accepted", it->name);
            if (snappy_validate_compressed_buffer(c, n - 1) !=
                SNAPPY_INVALID_INPUT)
                FAIL("%s: validate of a truncated stream", it->name);
            un = 0;
            if (snappy_uncompressed_length("\xff\xff\xff\xff\xff\xff", 6,
                                           &un) != SNAPPY_INVALID_INPUT)
                FAIL("bad varint accepted");
        }
        free(c);
        free(u);
    }
    printf("seam check ok: %d inputs\n", n_items);
    return 0;
}

struct job {
    struct item *it;
    int compress;
    double seconds;
    long calls;
    int bad;
    char *c, *u;
    size_t cn;
};

static void *worker(void *p)
{
    struct job *j = p;
    struct item *it = j->it;
    double t0 = now_s();
    for (;;) {
        if (j->compress) {
            size_t n = snappy_max_compressed_length(it->n_in);
            if (snappy_compress(it->in, it->n_in, j->c, &n) != SNAPPY_OK ||
                n != it->n_want)
                j->bad = 1;
        } else {
            size_t n = it->n_in;
            if (snappy_uncompress(it->want, it->n_want, j->u, &n) !=
                    SNAPPY_OK ||
                n != it->n_in)
                j->bad = 1;
        }
        j->calls++;
        if (now_s() - t0 >= j->seconds)
            break;
    }
    return NULL;
}

static int bench(int threads, double ms)
{
    if (threads < 1 || threads > 256)
        return 2;
    struct job *jobs = calloc(threads, sizeof *jobs);
    pthread_t *th = calloc(threads, sizeof *th);
    /* i = -1: an untimed round of all callers on the first file (a library
     * that makes its contexts on demand makes them here) */
    for (int i = -1; i < n_items; i++) {
        struct item *it = &items[i < 0 ? 0 : i];
        double rate[2];
        for (int dir = 0; dir < 2; dir++) {
            for (int t = 0; t < threads; t++) {
                jobs[t].it = it;
                jobs[t].compress = dir == 0;
                jobs[t].seconds = ms / 1e3;
                jobs[t].calls = 0;
                jobs[t].bad = 0;
                jobs[t].c = malloc(snappy_max_compressed_length(it->n_in));
                jobs[t].u = malloc(it->n_in + 1);
            }
            /* (one untimed call each way: contexts, staging) */
            size_t n = snappy_max_compressed_length(it->n_in);
            snappy_compress(it->in, it->n_in, jobs[0].c, &n);
            n = it->n_in;
            snappy_uncompress(it->want, it->n_want, jobs[0].u, &n);
            double t0 = now_s();
            for (int t = 0; t < threads; t++)
                pthread_create(&th[t], NULL, worker, &jobs[t]);
            long calls = 0;
            int bad = 0;
            for (int t = 0; t < threads; t++) {
                pthread_join(th[t], NULL);
                calls += jobs[t].calls;
                bad |= jobs[t].bad;
                free(jobs[t].c);
                free(jobs[t].u);
            }
            double dt = now_s() - t0;
            if (bad)
                FAIL("%s: a call failed under %d threads", it->name, threads);
            rate[dir] = calls * (double)it->n_in / dt / 1e6;
        }
        if (i < 0)
            continue;
        printf("%s %zu %d %.1f %.1f\n", it->name, it->n_in, threads, rate[0],
               rate[1]);
        fflush(stdout);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3 || load(argv[2])) {
        fprintf(stderr, "usage: seam_consumer check|bench <dir> [threads ms]\n");
        return 2;
    }
    if (!strcmp(argv[1], "check"))
        return check();
    if (!strcmp(argv[1], "bench") && argc >= 5)
        return bench(atoi(argv[3]), atof(argv[4]));
    return 2;
}
