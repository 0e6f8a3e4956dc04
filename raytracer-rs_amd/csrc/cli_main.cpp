// raytracer_mi355x: the reference's executable (main.rs:134-192) on the MI355X back end; see cli.cpp.
#include "../../include/rt_mi355x.h"

int main(int argc, char** argv) { return rtm_cli_main(argc, argv); }
