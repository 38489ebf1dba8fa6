// eleven_server.cpp -- the host process of the MI355X build: the reference's `ElevenRender` executable re-hosted
// without Boost / SYCL (reference src/main.cpp:190-240).  Listens on TCP port 5557 (the plug-in's port), serves one
// session at a time exactly like the reference's accept loop, and drives libeleven_hip.so through the C ABI.
//
//   eleven_server [--port N] [--loopback] [--once]
//     --port N      listen on N instead of 5557 (0 = any free port; the chosen port is printed)
//     --loopback    bind 127.0.0.1 instead of 0.0.0.0
//     --once        serve a single session and exit (tests)
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "eleven_commands.hpp"

int main(int argc, char** argv) {
    uint16_t port = 5557;
    bool loopback = false, once = false;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--port") && i + 1 < argc) port = (uint16_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--loopback")) loopback = true;
        else if (!strcmp(argv[i], "--once")) once = true;
        else { fprintf(stderr, "usage: eleven_server [--port N] [--loopback] [--once]\n"); return 2; }
    }
    signal(SIGPIPE, SIG_IGN);
    try {
        eleven::Acceptor acceptor(port, loopback);
        printf("listening on %u\n", (unsigned)acceptor.port);
        fflush(stdout);
        do {
            int fd = acceptor.accept_one();
            fprintf(stderr, "[eleven_server] connected\n");
            eleven::serve_session(fd);
            fprintf(stderr, "[eleven_server] disconnected\n");
        } while (!once);
    } catch (const std::exception& e) {
        fprintf(stderr, "[eleven_server] fatal: %s\n", e.what());
        return 1;
    }
    return 0;
}
