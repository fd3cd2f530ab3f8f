// io_probe — what bounds the end-to-end path on this box: pinned D2H rate and file-write rates on tmpfs.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/microbench/io_probe scripts/microbench/io_probe.cpp -lpthread
//   io_probe [dir=/dev/shm] [GB=8]
// Strategies for T writer threads, 32 MB slices taken round-robin from a pinned source buffer:
//   pwrite1   every thread pwrite()s its slices into ONE file          (inode lock)
//   pwriteN   thread t pwrite()s into its own file
//   mmap1     ONE file, ftruncate + one shared mapping, threads memcpy into it (page faults in parallel)
//   null      pwrite to /dev/null
//   mmaphp    mmap1 with madvise(MADV_HUGEPAGE) on the mapping (pays when /sys/kernel/mm/transparent_hugepage/shmem_enabled allows it)
//   prepw1    ONE file sized with ftruncate first (no write extends it), then pwrite1
//   fallocw1  ONE file allocated with fallocate first (timed), then pwrite1
//   cfr       the merge of sub-files: T threads copy_file_range() 16 source files into ONE pre-sized destination at their offsets
// io_probe sys prints what the box offers (kernel, mounts, THP settings) and returns.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/statvfs.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static void cat(const char *path) {
    FILE *f = fopen(path, "r");
    if (!f) { printf("%s: (absent)\n", path); return; }
    char buf[4096]; size_t n = fread(buf, 1, sizeof buf - 1, f); buf[n] = 0; fclose(f);
    printf("%s: %s%s", path, buf, n && buf[n - 1] == '\n' ? "" : "\n");
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "sys")) {
        cat("/proc/version"); cat("/sys/kernel/mm/transparent_hugepage/enabled"); cat("/sys/kernel/mm/transparent_hugepage/shmem_enabled");
        cat("/sys/kernel/mm/transparent_hugepage/hpage_pmd_size");
        FILE *f = fopen("/proc/mounts", "r");
        char line[1024];
        while (f && fgets(line, sizeof line, f))
            if (!strstr(line, "cgroup") && !strstr(line, " proc ") && !strstr(line, "sysfs") && !strstr(line, "devpts") && !strstr(line, "mqueue")) printf("mount: %s", line);
        if (f) fclose(f);
        return 0;
    }
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t total = (size_t)(argc > 2 ? atof(argv[2]) : 8.0) << 30;
    const size_t SL = 32u << 20, NS = 8;
    struct statvfs sv;
    if (!statvfs(dir.c_str(), &sv)) printf("%s: %.1f GB free; %ld cores online\n", dir.c_str(), (double)sv.f_bavail * sv.f_frsize / 1e9, sysconf(_SC_NPROCESSORS_ONLN));
    uint8_t *pin; void *dev;
    CK(hipHostMalloc((void **)&pin, SL * NS, hipHostMallocDefault));
    CK(hipMalloc(&dev, SL * NS));
    CK(hipMemset(dev, 'A', SL * NS));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (size_t sl : {(size_t)4 << 20, (size_t)16 << 20, SL, SL * NS}) {      // D2H into pinned memory, slices back to back on one stream
        CK(hipMemcpyAsync(pin, dev, sl, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        const double t0 = now(); size_t done = 0;
        while (done < total) { CK(hipMemcpyAsync(pin + (done % (SL * NS) / sl) * sl % (SL * NS), dev, sl, hipMemcpyDeviceToHost, st)); done += sl; }
        CK(hipStreamSynchronize(st));
        printf("D2H pinned, %3zu MB slices: %.1f GB/s\n", sl >> 20, (double)done / (now() - t0) / 1e9);
    }
    {   // two streams
        hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        const double t0 = now(); size_t done = 0;
        while (done < total) { CK(hipMemcpyAsync(pin, dev, SL, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(pin + SL, (uint8_t *)dev + SL, SL, hipMemcpyDeviceToHost, s2)); done += 2 * SL; }
        CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(s2));
        printf("D2H pinned, 2 streams x 32 MB: %.1f GB/s\n", (double)done / (now() - t0) / 1e9);
    }
    const size_t n_sl = total / SL;
    auto run = [&](const char *name, int T, int mode) {
        std::vector<int> fds;
        std::vector<std::string> paths;
        const int nf = mode == 1 ? T : 1;
        for (int f = 0; f < nf; ++f) {
            std::string p = mode == 3 ? "/dev/null" : dir + "/io_probe_" + std::to_string(getpid()) + "_" + std::to_string(f);
            int fd = open(p.c_str(), O_RDWR | O_CREAT | (mode == 3 ? 0 : O_TRUNC), 0644);
            if (fd < 0) { perror(p.c_str()); exit(2); }
            fds.push_back(fd); if (mode != 3) paths.push_back(p);
        }
        uint8_t *map = nullptr;
        std::vector<int> src_fds;
        if (mode == 7) {                    // sources of the merge: 16 files of total / 16 bytes (written here, untimed)
            for (int f = 0; f < 16; ++f) {
                std::string p = dir + "/io_probe_src_" + std::to_string(getpid()) + "_" + std::to_string(f);
                int fd = open(p.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
                if (fd < 0) { perror(p.c_str()); exit(2); }
                for (size_t d = 0; d < total / 16; d += SL) if (pwrite(fd, pin, SL, (off_t)d) != (ssize_t)SL) { perror("pwrite src"); exit(2); }
                src_fds.push_back(fd); paths.push_back(p);
            }
        }
        double t_pre = 0;
        if (mode == 6) { const double t = now(); if (posix_fallocate(fds[0], 0, (off_t)total)) { perror("fallocate"); } t_pre = now() - t; }
        const double t0 = now();
        if (mode == 2 || mode == 4 || mode == 5 || mode == 7) {
            if (ftruncate(fds[0], (off_t)total)) { perror("ftruncate"); exit(2); }
        }
        if (mode == 2 || mode == 4) {
            map = (uint8_t *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fds[0], 0);
            if (map == MAP_FAILED) { perror("mmap"); exit(2); }
            if (mode == 4 && madvise(map, total, MADV_HUGEPAGE)) perror("madvise(MADV_HUGEPAGE)");
        }
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n_sl) return;
                const uint8_t *src = pin + (i % NS) * SL;
                if (mode == 2 || mode == 4) { memcpy(map + i * SL, src, SL); continue; }
                if (mode == 7) {            // slice i of the destination comes from source file i / per_file
                    const size_t per_file = total / 16 / SL, f = i / per_file;
                    if (f >= 16) continue;
                    off_t in = (off_t)((i % per_file) * SL), out = (off_t)(i * SL); size_t d = 0;
                    while (d < SL) { ssize_t w = copy_file_range(src_fds[f], &in, fds[0], &out, SL - d, 0); if (w <= 0) { perror("copy_file_range"); exit(2); } d += (size_t)w; }
                    continue;
                }
                const int fd = fds[mode == 1 ? t : 0];
                const off_t off = mode == 1 ? (off_t)((i / T) * SL) : (off_t)(i * SL);
                size_t d = 0;
                while (d < SL) { ssize_t w = pwrite(fd, src + d, SL - d, off + (off_t)d); if (w <= 0) { perror("pwrite"); exit(2); } d += (size_t)w; }
            }
        });
        for (auto &x : th) x.join();
        if (map) munmap(map, total);
        const double dt = now() - t0;
        for (int fd : fds) close(fd);
        for (int fd : src_fds) close(fd);
        for (auto &p : paths) unlink(p.c_str());
        printf("%-8s T=%2d: %.1f GB/s", name, T, (double)(n_sl * SL) / dt / 1e9);
        if (mode == 6) printf("  (+ fallocate %.2f s = %.1f GB/s)", t_pre, (double)total / t_pre / 1e9);
        printf("\n");
        fflush(stdout);
    };
    for (int T : {1, 4, 8, 16, 32}) run("pwrite1", T, 0);
    for (int T : {4, 8, 16, 32}) run("pwriteN", T, 1);
    for (int T : {1, 4, 8, 16, 32, 64}) run("mmap1", T, 2);
    for (int T : {1, 8}) run("null", T, 3);
    for (int T : {1, 4, 16, 32}) run("mmaphp", T, 4);
    for (int T : {1, 8}) run("prepw1", T, 5);
    for (int T : {1, 8}) run("fallocw1", T, 6);
    for (int T : {1, 4, 16}) run("cfr", T, 7);
    return 0;
}
