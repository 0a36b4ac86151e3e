#include "bam-api/paired_reads.hpp"
