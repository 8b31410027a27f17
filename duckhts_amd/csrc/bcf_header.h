// bcf_header.h -- host-side BCF header dictionary + read_bcf schema ("column program") for the MI355X scan path.
//
// Replaces, on the bind path: htslib vcf.c bcf_hdr_parse 1410-1489, bcf_hdr_parse_line 653-789,
// bcf_hdr_register_hrec 831-1024 (shared FILTER/INFO/FORMAT dictionary, IDX= honoured, implicit PASS = first entry),
// bcf_hdr_parse_sample_line 286-314; and the schema construction of src/bcf_reader.c:540-760 with the VCF-spec
// Number corrections of src/include/vcf_types.h:46-224.  Pure host C++; the kernels see only the small tables built here.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace dhts {

enum { BCF_HL_FLT = 0, BCF_HL_INFO = 1, BCF_HL_FMT = 2 };
enum { BCF_HT_FLAG = 0, BCF_HT_INT = 1, BCF_HT_REAL = 2, BCF_HT_STR = 3 };
enum { BCF_VL_FIXED = 0, BCF_VL_VAR = 1, BCF_VL_A = 2, BCF_VL_G = 3, BCF_VL_R = 4 };

struct BcfDictEntry {
    bool present = false;                 // false = hole left by IDX= numbering
    std::string key;
    bool has[3] = {false, false, false};  // FILTER / INFO / FORMAT definition present
    int type[3] = {0, 0, 0};              // BCF_HT_*
    int vl[3] = {0, 0, 0};                // BCF_VL_* (only FIXED vs not matters to the schema)
    std::string info_desc;                // Description of the INFO definition (the VEP field list lives there)
};

struct BcfHeader {
    std::vector<BcfDictEntry> ids;        // BCF_DT_ID, index = dictionary id
    std::vector<std::string> ctg;         // BCF_DT_CTG, index = contig id
    std::vector<char> ctg_present;
    std::vector<int64_t> ctg_len;         // length= of the contig line (0 when absent): what the CSI writer sizes its binning by
    std::vector<std::string> samples;
    int version = 0;                      // major*1000000 + minor*1000, e.g. 4002000 (vcf.c:139-186)
    bool has_vep_tag = false;             // CSQ / BCSQ / ANN / VEP / vep INFO tag (vep_parser.c:100-118)
    int find_id(const std::string &k) const;
};

// returns false (and *err) when htslib's bcf_hdr_parse would fail
bool bcf_parse_header(const char *text, BcfHeader &h, std::string *err);

bool bcf_header_add_line(BcfHeader &h, const char *line);

// DuckDB logical type codes used by the schema (values of DUCKDB_TYPE_* in duckdb.h)
enum { DT_BOOLEAN = 1, DT_INTEGER = 4, DT_BIGINT = 5, DT_FLOAT = 10, DT_DOUBLE = 11, DT_VARCHAR = 17 };

// kinds of output column
enum { BK_CHROM = 0, BK_POS, BK_ID, BK_REF, BK_ALT, BK_QUAL, BK_FILTER, BK_INFO, BK_SAMPLE_ID, BK_FORMAT,
       BK_VEP };       // VEP_<field>: LIST per transcript out of the annotation tag's INFO string (field = index of the field inside a transcript)

struct VepField { std::string name; int htype = BCF_HT_STR; };

struct BcfField { std::string name; int id = -1; int htype = BCF_HT_STR; bool is_list = false; };

struct BcfColumn {
    std::string name;
    int kind = 0;
    int duck_type = DT_VARCHAR;           // element type
    bool is_list = false;
    int field = -1;                       // index into info_fields / format_fields
    int sample = -1;                      // wide FORMAT: sample index; tidy FORMAT: -1 (row % n_samples)
};

struct BcfSchema {
    std::vector<BcfField> info_fields, format_fields;
    std::vector<BcfColumn> cols;
    bool tidy = false;                    // tidy_format && n_samples > 0 (bcf_reader.c:1192)
    int n_samples = 0;
    bool gt_string_ok = false;            // header has FORMAT/GT declared String: the GT getter works (vcf.c:6183-6187)
    int gt_id = -1;                       // dictionary id of "GT" (any class), -1 if absent: updatephasing key (vcf.c:2063-2067)
    // VEP / BCSQ / ANN annotation (src/vep_parser.c:100-182, src/bcf_reader.c:582-603): the tag, its '|'-separated fields, and the
    // entry of info_fields that is the tag itself (the cells kernel reads the tag's INFO string through that slot)
    std::string vep_tag; std::vector<VepField> vep_fields; int vep_info_field = -1;
};

void bcf_build_schema(const BcfHeader &h, bool tidy_format, BcfSchema &s);

}  // namespace dhts
