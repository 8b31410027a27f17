/* bcf_oracle.c -- CPU restatement of the read_bcf scan path (placeholder until the BCF row lands). */
#include "dhts_oracle.h"
