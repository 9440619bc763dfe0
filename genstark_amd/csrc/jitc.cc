// gstark_jitc — the compiler of AIR programs as a process of its own (air_jit.hip: background builds of the default "auto" mode).
//
// A background build inside the proving process means a thread inside hiprtc / comgr (LLVM) for seconds.  If the host reaches exit()
// meanwhile, the C++ runtime destroys that compiler's statics under the thread: seen on the GPU box as "LLVM ERROR ..." + abort at
// exit, and once as a process that never ended.  No ordering of atexit handlers closes that race (the compiler registers finalisers
// whenever one of its lazily built objects first appears).  So the proving process never runs a background compilation itself: it
// hands the generated source to this helper over a socket and waits for the code object; when the host exits first the helper's
// reply has nowhere to go and the helper ends — after having written the code object to the cache, so the work is not lost.
//
// stdin = stdout = one end of a socketpair.
//   request  "GSJ1" u32 nitems { u32 name_len, name, u64 len, bytes }*  u32 path_len, cache path ("" = none)
//            item 0: name = kernel entry point, bytes = generated source; items 1..: the field headers the source includes
//   reply    "GSOK" u64 len, code object   |   "GSER" u64 len, compiler log
// Needs no GPU (hiprtc cross-compiles for gfx950).
#include <hip/hiprtc.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <vector>

static bool read_all(void *dst, size_t n) {
    char *p = (char *)dst;
    while (n) {
        const ssize_t r = read(0, p, n);
        if (r <= 0) return false;
        p += r;
        n -= (size_t)r;
    }
    return true;
}
static bool write_all(const void *src, size_t n) {
    const char *p = (const char *)src;
    while (n) {
        ssize_t r = send(1, p, n, MSG_NOSIGNAL);
        if (r < 0) r = write(1, p, n);          // stdout is not a socket (run by hand): plain write
        if (r <= 0) return false;
        p += r;
        n -= (size_t)r;
    }
    return true;
}
static void reply(const char *tag, const void *data, uint64_t len) {
    write_all(tag, 4);
    write_all(&len, 8);
    write_all(data, len);
}
// same rules as air_jit.hip's jit_disk_write: a new 0600 file of this user, renamed into place
static void cache_write(const std::string &path, const std::vector<char> &code) {
    if (path.empty() || code.empty()) return;
    char tmp[32];
    snprintf(tmp, sizeof tmp, ".%d.tmp", (int)getpid());
    const std::string t = path + tmp;
    const int fd = open(t.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return;
    size_t put = 0;
    while (put < code.size()) {
        const ssize_t r = write(fd, code.data() + put, code.size() - put);
        if (r <= 0) break;
        put += (size_t)r;
    }
    close(fd);
    if (put == code.size()) rename(t.c_str(), path.c_str()); else remove(t.c_str());
}

int main() {
    char magic[4];
    uint32_t nitems = 0;
    if (!read_all(magic, 4) || memcmp(magic, "GSJ1", 4) != 0 || !read_all(&nitems, 4) || nitems < 1 || nitems > 8) return 2;
    std::vector<std::string> names(nitems), bodies(nitems);
    for (uint32_t i = 0; i < nitems; i++) {
        uint32_t nl = 0;
        uint64_t dl = 0;
        if (!read_all(&nl, 4) || nl > 256) return 2;
        names[i].resize(nl);
        if (!read_all(&names[i][0], nl) || !read_all(&dl, 8) || dl > (64ull << 20)) return 2;
        bodies[i].resize(dl);
        if (!read_all(&bodies[i][0], dl)) return 2;
    }
    uint32_t pl = 0;
    if (!read_all(&pl, 4) || pl > 4096) return 2;
    std::string path(pl, 0);
    if (pl && !read_all(&path[0], pl)) return 2;

    std::vector<const char *> hn, hb;
    for (uint32_t i = 1; i < nitems; i++) { hn.push_back(names[i].c_str()); hb.push_back(bodies[i].c_str()); }
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, bodies[0].c_str(), "gs_air_jit.hip", (int)hn.size(), hb.data(), hn.data()) != HIPRTC_SUCCESS) {
        const char *m = "hiprtcCreateProgram failed";
        reply("GSER", m, strlen(m));
        return 1;
    }
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    if (hiprtcCompileProgram(prog, 3, opts) != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, 0);
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        reply("GSER", log.data(), log.size());
        return 1;
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    hiprtcGetCode(prog, code.data());
    cache_write(path, code);                    // first: the host may be gone by now, the next process still finds the program
    reply("GSOK", code.data(), code.size());
    return 0;
}
