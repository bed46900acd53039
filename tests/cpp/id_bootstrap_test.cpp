// id_bootstrap_test.cpp -- examples/id_bootstrap.hpp (the rendezvous of examples/tiled_host.cpp) on the CPU:
// stale files of other sessions are ignored, readers wait for their own session's file, publication is atomic.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../examples/id_bootstrap.hpp"

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main() {
  using namespace rna_bootstrap;
  char dir[] = "/tmp/rna_idtestXXXXXX";
  CHECK(mkdtemp(dir) != nullptr);
  const std::string file = std::string(dir) + "/id";
  unsigned char stale[128], fresh[128], got[128];
  for (int k = 0; k < 128; ++k) { stale[k] = (unsigned char)(k * 7 + 1); fresh[k] = (unsigned char)(255 - k); }

  // tokens: explicit ones differ per text, the default one is the same for every child of this launcher
  const uint64_t t_old = session_token("job-41"), t_new = session_token("job-42");
  CHECK(t_old != t_new && t_old != 0 && t_new != 0);
  CHECK(session_token(nullptr) == session_token("") && session_token(nullptr) != 0);

  // 1. nothing there
  CHECK(try_fetch(file.c_str(), t_new, got, 128) == 0);
  // 2. an earlier job's file: seen, refused, and a reader of the new job times out on it instead of taking it
  CHECK(publish(file.c_str(), t_old, stale, 128));
  CHECK(try_fetch(file.c_str(), t_new, got, 128) == -1);
  CHECK(!fetch(file.c_str(), t_new, got, 128, 0.3));
  CHECK(try_fetch(file.c_str(), t_old, got, 128) == 1 && std::memcmp(got, stale, 128) == 0);
  // 3. round 3's file format (the bare 128 bytes, no token) is refused as well
  {
    FILE* f = std::fopen(file.c_str(), "wb");
    CHECK(f && std::fwrite(stale, 128, 1, f) == 1);
    std::fclose(f);
    CHECK(try_fetch(file.c_str(), t_new, got, 128) != 1);
  }
  // 4. readers that started BEFORE rank 0 published (the stale file still in place) get the fresh id
  CHECK(publish(file.c_str(), t_old, stale, 128));
  std::vector<std::thread> readers;
  int ok[4] = {0, 0, 0, 0};
  for (int r = 0; r < 4; ++r)
    readers.emplace_back([&, r] {
      unsigned char mine[128];
      ok[r] = fetch(file.c_str(), t_new, mine, 128, 10.0) && std::memcmp(mine, fresh, 128) == 0;
    });
  std::this_thread::sleep_for(std::chrono::milliseconds(300));
  CHECK(publish(file.c_str(), t_new, fresh, 128));
  for (auto& t : readers) t.join();
  CHECK(ok[0] && ok[1] && ok[2] && ok[3]);
  // 5. a half-written file never appears under the final name: hammer publish against try_fetch
  {
    bool bad = false, stop = false;
    std::thread w([&] { for (int k = 0; k < 2000; ++k) publish(file.c_str(), t_new, (k & 1) ? fresh : stale, 128); stop = true; });
    while (!stop) {
      unsigned char m[128];
      const int rc = try_fetch(file.c_str(), t_new, m, 128);
      if (rc == 1 && std::memcmp(m, fresh, 128) != 0 && std::memcmp(m, stale, 128) != 0) bad = true;
      if (rc == -1) bad = true;
    }
    w.join();
    CHECK(!bad);
  }
  std::remove(file.c_str());
  rmdir(dir);
  std::printf("id bootstrap ok\n");
  return 0;
}
