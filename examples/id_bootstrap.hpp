// id_bootstrap.hpp -- the rendezvous of examples/tiled_host.cpp: rank 0 hands an opaque id (RCCL's ncclUniqueId)
// to the other ranks of ITS job through a file in a directory they share.  No HIP, no RCCL here: tests/cpp/
// id_bootstrap_test.cpp runs it on the CPU.
//
// A file left behind by an earlier job must never be taken for this job's (the id inside names a communicator that no
// longer exists: ncclCommInitRank would wait for ever).  Every file therefore carries a SESSION token and a reader only
// accepts the token of its own job.  The launcher passes one, fresh per job (argv / RNA_TILED_SESSION); a job of MORE THAN
// ONE rank must (examples/tiled_host.cpp refuses to start without it).  The default -- (parent pid, parent start time) --
// only tells jobs of DIFFERENT launchers apart: two jobs started one after the other by the same long-lived shell, pytest
// process or job script share it, a rank > 0 of the second job that starts before its rank 0 has re-published would
// accept the first job's id, and ncclCommInitRank would hang -- so the default serves single-rank runs only, where nobody
// reads the file.
// The writer publishes with write-to-temporary + rename(), so a reader never sees half a file.
#pragma once

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include <unistd.h>

namespace rna_bootstrap {

constexpr uint64_t MAGIC = 0x31444941'4e52ull;   // "RNAID1"

// session token of this process's job: `given` if the launcher passed one, else (parent pid, parent start time)
inline uint64_t session_token(const char* given) {
  if (given && *given) {
    uint64_t h = 1469598103934665603ull;   // FNV-1a over the string: any launcher-chosen text will do
    for (const char* c = given; *c; ++c) h = (h ^ (unsigned char)*c) * 1099511628211ull;
    return h | 1ull;
  }
  const long ppid = (long)getppid();
  unsigned long long start = 0;
  char path[64];
  std::snprintf(path, sizeof(path), "/proc/%ld/stat", ppid);
  if (FILE* f = std::fopen(path, "r")) {
    char buf[1024];
    const size_t n = std::fread(buf, 1, sizeof(buf) - 1, f);
    std::fclose(f);
    buf[n] = 0;
    // field 22 (starttime), counted after the last ')' of the command name
    if (const char* p = std::strrchr(buf, ')')) {
      int field = 2;
      for (++p; *p && field < 22; ++p)
        if (*p == ' ') ++field;
      start = std::strtoull(p, nullptr, 10);
    }
  }
  return (((uint64_t)ppid << 40) ^ (uint64_t)start) | 1ull;
}

// rank 0: publish `bytes` of id for session `token` (atomically replaces whatever the file held)
inline bool publish(const char* file, uint64_t token, const void* id, size_t bytes) {
  const std::string tmp = std::string(file) + ".tmp." + std::to_string((long)getpid());
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const uint64_t head[2] = {MAGIC, token};
  const bool ok = std::fwrite(head, sizeof(head), 1, f) == 1 && std::fwrite(id, bytes, 1, f) == 1;
  if (std::fclose(f) != 0 || !ok) { std::remove(tmp.c_str()); return false; }
  if (std::rename(tmp.c_str(), file) != 0) { std::remove(tmp.c_str()); return false; }
  return true;
}

// 1: read, 0: no file of this session (yet), -1: a file is there but belongs to another session (stale)
inline int try_fetch(const char* file, uint64_t token, void* id, size_t bytes) {
  FILE* f = std::fopen(file, "rb");
  if (!f) return 0;
  uint64_t head[2] = {0, 0};
  const bool ok = std::fread(head, sizeof(head), 1, f) == 1 && std::fread(id, bytes, 1, f) == 1;
  std::fclose(f);
  if (!ok) return 0;
  if (head[0] != MAGIC || head[1] != token) return -1;
  return 1;
}

// the other ranks: wait (at most timeout_s) for rank 0's file of THIS session; stale files are ignored
inline bool fetch(const char* file, uint64_t token, void* id, size_t bytes, double timeout_s) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    if (try_fetch(file, token, id, bytes) == 1) return true;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
  }
}

}  // namespace rna_bootstrap
