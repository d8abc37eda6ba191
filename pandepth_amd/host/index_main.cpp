// index_main.cpp — `pandepth_index in.bam`: writes in.bam.bai (see bai_build in bam.h).
#include <iostream>
#include "bam.h"

int main(int argc, char **argv)
{
    if (argc != 2) { std::cerr << "usage: pandepth_index in.bam   (writes in.bam.bai)" << std::endl; return 2; }
    std::string err;
    if (!pdh::bai_build(argv[1], &err)) { std::cerr << "Error: " << err << std::endl; return 1; }
    return 0;
}
