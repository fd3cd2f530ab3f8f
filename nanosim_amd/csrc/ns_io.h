// ns_io.h — the output side of a worker (host code only): result buffers leave HBM for files WHILE the next batch is generated.
//
// The reference's workers write every record and every error-profile row as they go (out_reads.write S:1437-1443, out_error.write
// S:2006-2008).  Here a batch's FASTA/FASTQ image and error-profile image are complete in HBM when ns_generate returns; ns_sink_write
// queues them for a file and returns at once:
//   copier thread   cuts the buffer into slices, takes a free page-locked staging slice, issues hipMemcpyAsync on the context's COPY
//                   stream (its own HIP stream: DMA engines, no kernel) and records an event behind it; several slices are in flight
//   writer threads  wait for a slice's event, pwrite() the bytes at their final file offset, hand the staging slice back
// The context keeps TWO result slots (record + error-profile buffers): while slot s is being copied out, the next ns_generate fills
// slot s ^ 1; the one after that waits until the copies out of s have left the device (not until they are in the file).
#pragma once
#include <hip/hip_runtime.h>
#include <errno.h>
#include <string.h>
#include <unistd.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct ns_sink {
    int fd = -1;                          // -1: the bytes are copied to the host and dropped (measures the device-to-host rate alone)
    uint64_t off = 0;                     // file offset of the next byte queued
    std::atomic<uint64_t> queued{0}, written{0};
    std::atomic<int> err{0};              // errno of the first failed write
};

struct IoEngine {
    int device = 0;
    hipStream_t stream = nullptr;
    size_t slice_bytes = 0;
    struct Slice { uint8_t *pin = nullptr; hipEvent_t t0 = nullptr, t1 = nullptr; };
    std::vector<Slice> slices;
    struct CopyJob { const uint8_t *src; uint64_t n; ns_sink *sink; uint64_t file_off; int slot; };
    struct WriteJob { int slice; ns_sink *sink; uint64_t file_off; size_t n; int slot; };
    std::mutex mu;
    std::condition_variable cv_free, cv_copy, cv_write, cv_idle;
    std::deque<int> free_slices;
    std::deque<CopyJob> copy_jobs;
    std::deque<WriteJob> write_jobs;
    uint64_t slot_pending[2] = {0, 0};    // slices of the slot that have not left the device yet
    uint64_t jobs_open = 0;               // slices queued and not yet written
    bool stop = false;
    std::string err;                      // first HIP error of the copier / writers
    std::thread copier;
    std::vector<std::thread> writers;
    // accounting (ns_io_counters): bytes and DMA time of the slices, time the copier waited for a free staging slice
    double dma_ms = 0, wait_free_s = 0, write_s = 0;
    uint64_t bytes = 0;

    static size_t env_or(const char *name, size_t dflt) { const char *v = getenv(name); return v && atoll(v) > 0 ? (size_t)atoll(v) : dflt; }

    int start(int dev, std::string &msg) {
        device = dev;
        slice_bytes = env_or("NS_IO_SLICE_BYTES", env_or("NS_IO_SLICE_MB", 16) << 20);      // (bytes: tests that want many slices per small batch)
        const size_t n_slices = env_or("NS_IO_SLICES", 16), n_threads = env_or("NS_IO_THREADS", 8);
        if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) { msg = "ns_io: no copy stream"; return -1; }
        slices.resize(n_slices);
        for (size_t i = 0; i < n_slices; ++i) {
            Slice &s = slices[i];
            if (hipHostMalloc((void **)&s.pin, slice_bytes, hipHostMallocDefault) != hipSuccess || hipEventCreate(&s.t0) != hipSuccess ||
                hipEventCreate(&s.t1) != hipSuccess) { msg = "ns_io: staging allocation failed"; return -1; }
            free_slices.push_back((int)i);
        }
        copier = std::thread([this] { copy_loop(); });
        for (size_t t = 0; t < n_threads; ++t) writers.emplace_back([this] { write_loop(); });
        return 0;
    }
    void shutdown() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv_copy.notify_all(); cv_write.notify_all(); cv_free.notify_all();
        if (copier.joinable()) copier.join();
        for (auto &t : writers) if (t.joinable()) t.join();
        hipError_t e = hipSetDevice(device); (void)e;
        if (stream) { e = hipStreamSynchronize(stream); e = hipStreamDestroy(stream); }
        for (Slice &s : slices) {
            if (s.pin) e = hipHostFree(s.pin);
            if (s.t0) e = hipEventDestroy(s.t0);
            if (s.t1) e = hipEventDestroy(s.t1);
        }
        slices.clear();
    }
    void fail(const std::string &m) { std::lock_guard<std::mutex> g(mu); if (err.empty()) err = m; }

    // queue bytes [0, n) of a device buffer of result slot `slot` for sink `s`
    void enqueue(ns_sink *s, const void *src, uint64_t n, int slot) {
        if (!n) return;
        const uint64_t n_sl = (n + slice_bytes - 1) / slice_bytes;
        {
            std::lock_guard<std::mutex> g(mu);
            copy_jobs.push_back(CopyJob{static_cast<const uint8_t *>(src), n, s, s->off, slot});
            slot_pending[slot] += n_sl; jobs_open += n_sl;
        }
        s->off += n; s->queued += n;
        cv_copy.notify_one();
    }
    void wait_slot(int slot) { std::unique_lock<std::mutex> g(mu); cv_idle.wait(g, [&] { return slot_pending[slot] == 0; }); }
    bool slot_busy(int slot) { std::lock_guard<std::mutex> g(mu); return slot_pending[slot] != 0; }
    void wait_all() { std::unique_lock<std::mutex> g(mu); cv_idle.wait(g, [&] { return jobs_open == 0; }); }
    void wait_sink(ns_sink *s) { std::unique_lock<std::mutex> g(mu); cv_idle.wait(g, [&] { return s->written.load() == s->queued.load(); }); }

    void copy_loop() {
        hipError_t e = hipSetDevice(device); (void)e;
        for (;;) {
            CopyJob j;
            {
                std::unique_lock<std::mutex> g(mu);
                cv_copy.wait(g, [&] { return stop || !copy_jobs.empty(); });
                if (copy_jobs.empty()) return;
                j = copy_jobs.front(); copy_jobs.pop_front();
            }
            for (uint64_t pos = 0; pos < j.n; pos += slice_bytes) {
                const size_t n = (size_t)std::min<uint64_t>(slice_bytes, j.n - pos);
                int si;
                {
                    std::unique_lock<std::mutex> g(mu);
                    const auto t0 = std::chrono::steady_clock::now();
                    cv_free.wait(g, [&] { return stop || !free_slices.empty(); });
                    wait_free_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    if (free_slices.empty()) return;
                    si = free_slices.front(); free_slices.pop_front();
                }
                Slice &s = slices[si];
                hipError_t e1 = hipEventRecord(s.t0, stream);
                hipError_t e2 = hipMemcpyAsync(s.pin, j.src + pos, n, hipMemcpyDeviceToHost, stream);
                hipError_t e3 = hipEventRecord(s.t1, stream);
                if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)
                    fail(std::string("ns_io: device-to-host copy: ") + hipGetErrorString(e2 != hipSuccess ? e2 : e1 != hipSuccess ? e1 : e3));
                {
                    std::lock_guard<std::mutex> g(mu);
                    write_jobs.push_back(WriteJob{si, j.sink, j.file_off + pos, n, j.slot});
                }
                cv_write.notify_one();
            }
        }
    }
    void write_loop() {
        hipError_t e = hipSetDevice(device); (void)e;
        for (;;) {
            WriteJob j;
            {
                std::unique_lock<std::mutex> g(mu);
                cv_write.wait(g, [&] { return stop || !write_jobs.empty(); });
                if (write_jobs.empty()) return;
                j = write_jobs.front(); write_jobs.pop_front();
            }
            Slice &s = slices[j.slice];
            float ms = 0;
            hipError_t e1 = hipEventSynchronize(s.t1);
            if (e1 == hipSuccess) e1 = hipEventElapsedTime(&ms, s.t0, s.t1);
            if (e1 != hipSuccess) fail(std::string("ns_io: copy event: ") + hipGetErrorString(e1));
            {
                std::lock_guard<std::mutex> g(mu);
                --slot_pending[j.slot]; dma_ms += ms; bytes += j.n;
            }
            cv_idle.notify_all();
            const auto t0 = std::chrono::steady_clock::now();
            if (j.sink->fd >= 0 && !j.sink->err.load()) {
                size_t done = 0;
                while (done < j.n) {
                    const ssize_t w = pwrite(j.sink->fd, s.pin + done, j.n - done, (off_t)(j.file_off + done));
                    if (w < 0) { if (errno == EINTR) continue; int z = 0; j.sink->err.compare_exchange_strong(z, errno ? errno : EIO); break; }
                    if (w == 0) { int z = 0; j.sink->err.compare_exchange_strong(z, ENOSPC); break; }
                    done += (size_t)w;
                }
            }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            j.sink->written += j.n;
            {
                std::lock_guard<std::mutex> g(mu);
                free_slices.push_back(j.slice); --jobs_open; write_s += dt;
            }
            cv_free.notify_one(); cv_idle.notify_all();
        }
    }
};
