#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "linearise.hpp"
extern "C" {
int dr_scene_load(const char*, const char*, dr_scene**);
void dr_scene_free(dr_scene*);
int dr_scene_build_bvh(dr_scene*, int);
int dr_scene_save_binary(const dr_scene*, const char*);
int dr_scene_load_binary(const char*, dr_scene**);
const char* dr_last_error(void);
}
using namespace dr;
int main(int argc, char** argv) {
  int bad = 0;
  for (int i = 2; i < argc; i++) {
    dr_scene* s = nullptr;
    int rc = dr_scene_load(argv[i], argv[1], &s);
    if (rc != 0) { printf("%s: load rc %d (%s)\n", argv[i], rc, dr_last_error()); continue; }
    rc = dr_scene_build_bvh(s, 3);
    if (rc != 0) { printf("%s: build rc %d (%s)\n", argv[i], rc, dr_last_error()); dr_scene_free(s); continue; }
    DeviceImage img;
    rc = linearise(s->host, img);
    if (rc != 0) { printf("%s: linearise rc %d (%s)\n", argv[i], rc, dr_last_error()); bad++; }
    rc = dr_scene_save_binary(s, "/tmp/dogeray_asan_x.rtsb");
    dr_scene* t = nullptr;
    if (rc == 0 && dr_scene_load_binary("/tmp/dogeray_asan_x.rtsb", &t) == 0) {
      if (t->host.objects.size() != s->host.objects.size() || memcmp(t->host.bvh.data(), s->host.bvh.data(), s->host.bvh.size() * sizeof(dr_bvh_node)) != 0) { printf("%s: rtsb mismatch\n", argv[i]); bad++; }
      dr_scene_free(t);
    } else { printf("%s: rtsb rc %d (%s)\n", argv[i], rc, dr_last_error()); bad++; }
    // corrupted links must be refused, never followed (miss_node indexed new_id[] unchecked)
    {
      const size_t used = (size_t)s->host.bvh_used, all = s->host.bvh.size();
      struct Mut { size_t node; int field; int value; };
      const Mut muts[] = {{0, 0, (int)all + 7}, {used / 2, 1, -5}, {used - 1, 1, 1 << 30}, {0, 2, -3}, {used / 3, 3, (int)all}, {1, 0, -2}};
      for (const Mut& m : muts) {
        dr_scene bad_scene;
        bad_scene.host = s->host;
        dr_bvh_node& b = bad_scene.host.bvh[m.node < all ? m.node : 0];
        if (m.field == 0) b.hit_node = m.value; else if (m.field == 1) b.miss_node = m.value; else if (m.field == 2) b.children[0] = m.value; else b.children[1] = m.value;
        DeviceImage junk;
        const bool touches = b.active && (m.field < 2 || !b.end);
        if (linearise(bad_scene.host, junk) == 0 && touches) { printf("%s: corrupted link accepted (node %zu field %d)\n", argv[i], m.node, m.field); bad++; }
      }
    }
    printf("%s: ok, %d objects, walk %zu units, wide %zu records depth %d\n", argv[i], s->host.n, img.walk.size(), img.wide.size() / 4, img.wide_depth);
    dr_scene_free(s);
  }
  return bad;
}
