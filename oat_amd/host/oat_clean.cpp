// oat-clean-hip NAME...   (src/cleaner/main.cpp:122-155): unlink "<NAME>_node" and "<NAME>_obj".
#include "shmemdf.hpp"
#include <cstdio>
namespace oat { volatile sig_atomic_t quit = 0; }
int main(int argc, char **argv)
{
    for (int i = 1; i < argc; ++i) {
        std::string n = argv[i];
        bool a = oat::remove_segment(n + "_node"), b = oat::remove_segment(n + "_obj");
        printf("%s: %s\n", argv[i], (a || b) ? "removed" : "nothing to remove");
    }
    return 0;
}
