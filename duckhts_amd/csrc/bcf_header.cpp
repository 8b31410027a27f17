// bcf_header.cpp -- see bcf_header.h.  Host-only; no htslib, no oracle code.
#include "bcf_header.h"

#include <ctype.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

namespace dhts {

namespace {

struct HeaderLine {
    std::string key;                      // text between "##" and '='
    bool structured = false;              // value starts with '<'
    std::string value;                    // generic lines only
    std::vector<std::pair<std::string, std::string>> kv;
    const std::string *get(const char *k, bool nocase) const {
        for (auto &p : kv) if (nocase ? !strcasecmp(p.first.c_str(), k) : p.first == k) return &p.second;
        return nullptr;
    }
};

bool alpha(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
bool alnum(char c) { return alpha(c) || (c >= '0' && c <= '9'); }

// Tokeniser for one "##" line with the quoting / bracket / nesting rules of vcf.c:653-789.
// Returns 1 = parsed, 0 = not a ## line (len 0) or malformed line (len > 0, skipped), -1 = fatal.
int parse_line(const char *line, HeaderLine &out, size_t &len) {
    out = HeaderLine();
    len = 0;
    if (line[0] != '#' || line[1] != '#') return 0;
    const char *p = line + 2, *q = p;
    auto eol_len = [&](const char *e) { while (*e && *e != '\n') e++; return (size_t)(e - line) + (*e ? 1 : 0); };
    while (*q && *q != '=' && *q != '\n') q++;
    if (*q != '=' || q == p) { len = eol_len(q); return 0; }
    out.key.assign(p, q);
    p = ++q;
    if (*p != '<') {
        while (*q && *q != '\n') q++;
        out.value.assign(p, q);
        len = (size_t)(q - line) + (*q ? 1 : 0);
        return 1;
    }
    out.structured = true;
    int depth = 1;
    while (*q && *q != '\n' && depth > 0) {
        p = ++q;
        while (*q == ' ') { p++; q++; }
        if (p == q && *q && (alpha(*q) || *q == '_')) { q++; while (*q && (alnum(*q) || *q == '_' || *q == '.')) q++; }
        const char *kend = q;
        while (*q == ' ') q++;
        if (*q != '=' || kend == p) { len = eol_len(q); return 0; }
        std::string key(p, kend);
        p = ++q;
        while (*q == ' ') { p++; q++; }
        bool quoted = false; char closer = 0;
        if (*p == '"') { quoted = true; closer = '"'; p++; }
        else if (*p == '[') { quoted = true; closer = ']'; }
        if (quoted) q++;
        for (; *q && *q != '\n'; q++) {
            if (quoted) {
                if (*q == closer) { int bs = 0; for (const char *b = q - 1; b >= p && *b == '\\'; b--) bs++; if (bs % 2 == 0) break; }
            } else {
                if (*q == '<') depth++;
                if (*q == '>') depth--;
                if (!depth) break;
                if (*q == ',' && depth == 1) break;
            }
        }
        const char *vend = q;
        if (quoted && closer == ']') {
            if (*q != closer) return -1;
            vend++; q++; quoted = false;
        }
        while (vend > p && vend[-1] == ' ') vend--;
        out.kv.emplace_back(key, std::string(p, vend));
        if (quoted && *q == closer) q++;
        if (*q == '>') { if (depth) depth--; q++; }
    }
    len = eol_len(q);
    return 1;
}

bool parse_idx(const std::string &s, int &idx) {
    char *end = nullptr; long v = strtol(s.c_str(), &end, 10);
    if (*end || v < 0 || v >= (long)INT_MAX - 1) return false;
    idx = (int)v; return true;
}

// -1 fatal, otherwise 0/1 like bcf_hdr_register_hrec
int register_line(BcfHeader &h, const HeaderLine &r) {
    int hl;
    if (r.key == "contig") hl = 3;
    else if (r.key == "INFO") hl = BCF_HL_INFO;
    else if (r.key == "FILTER") hl = BCF_HL_FLT;
    else if (r.key == "FORMAT") hl = BCF_HL_FMT;
    else return 0;
    if (!r.structured) return 0;
    if (hl == 3) {
        long long ctg_length = 0;
        if (const std::string *l = r.get("length", true)) {
            char *end = nullptr; long long v = strtoll(l->c_str(), &end, 10);
            if (end == l->c_str() || v < 0) return 0;
            ctg_length = v;
        }
        const std::string *id = r.get("ID", true);
        if (!id) return 0;
        for (size_t i = 0; i < h.ctg.size(); i++) if (h.ctg_present[i] && h.ctg[i] == *id) return 0;
        int idx = -1;
        if (const std::string *x = r.get("IDX", true)) if (!parse_idx(*x, idx)) return 0;
        if (idx == -1) idx = (int)h.ctg.size();
        else if (idx < (int)h.ctg.size() && h.ctg_present[idx]) return -1;      // conflicting IDX (vcf.c:803-809)
        if (idx >= (int)h.ctg.size()) { h.ctg.resize(idx + 1); h.ctg_present.resize(idx + 1, 0); }
        if ((int)h.ctg_len.size() < (int)h.ctg.size()) h.ctg_len.resize(h.ctg.size(), 0);
        h.ctg[idx] = *id; h.ctg_present[idx] = 1; h.ctg_len[idx] = ctg_length;
        return 1;
    }
    const std::string *id = nullptr, *desc = nullptr; int type = -1, var = -1, num = -1, idx = -1;
    for (auto &p : r.kv) {
        if (p.first == "ID") id = &p.second;
        else if (p.first == "Description") { if (!desc) desc = &p.second; }
        else if (p.first == "IDX") { if (!parse_idx(p.second, idx)) return 0; }
        else if (p.first == "Type") {
            const std::string &v = p.second;
            type = v == "Integer" ? BCF_HT_INT : v == "Float" ? BCF_HT_REAL : v == "Flag" ? BCF_HT_FLAG : BCF_HT_STR;
        } else if (p.first == "Number") {
            const std::string &v = p.second; const bool fmt = hl == BCF_HL_FMT;
            if (v == "A") var = BCF_VL_A; else if (v == "R") var = BCF_VL_R; else if (v == "G") var = BCF_VL_G; else if (v == ".") var = BCF_VL_VAR;
            else if (fmt && (v == "P" || v == "LA" || v == "LR" || v == "LG" || v == "M")) var = 5;          // non-fixed classes of VCF >= 4.4
            else if (sscanf(v.c_str(), "%d", &num) == 1) var = BCF_VL_FIXED;
            if (var != BCF_VL_FIXED) num = 0xfffff;
        }
    }
    if (hl == BCF_HL_INFO || hl == BCF_HL_FMT) {
        if (type == -1) type = BCF_HT_STR;
        if (var == -1) var = BCF_VL_VAR;
        if (type == BCF_HT_FLAG && (var != BCF_VL_FIXED || num != 0)) { var = BCF_VL_FIXED; num = 0; }
    }
    if (!id) return 0;
    int k = h.find_id(*id);
    if (k < 0) {
        k = idx;
        if (k == -1) k = (int)h.ids.size();
        else if (k < (int)h.ids.size() && h.ids[k].present) return -1;
        if (k >= (int)h.ids.size()) h.ids.resize(k + 1);
        h.ids[k].present = true; h.ids[k].key = *id;
    } else if (h.ids[k].has[hl]) return 0;
    h.ids[k].has[hl] = true; h.ids[k].type[hl] = type & 0xf; h.ids[k].vl[hl] = var & 0xf;
    if (hl == BCF_HL_INFO && desc) h.ids[k].info_desc = *desc;
    return 1;
}

int version_of(const std::string &v) {
    size_t a = v.find("VCFv");
    if (a == std::string::npos) return 4002000;
    size_t dot = v.find('.', a + 4);
    if (dot == std::string::npos) return 4002000;
    return (int)(strtol(v.c_str() + a + 4, nullptr, 10) * 1000000 + strtol(v.c_str() + dot + 1, nullptr, 10) * 1000);
}

struct Spec { const char *name; int vl; };
// reserved keys whose Number the reader corrects to the VCF specification (vcf_types.h:46-93)
const Spec kFmtSpec[] = {{"AD", BCF_VL_R}, {"ADF", BCF_VL_R}, {"ADR", BCF_VL_R}, {"EC", BCF_VL_A}, {"GL", BCF_VL_G}, {"GP", BCF_VL_G}, {"PL", BCF_VL_G},
    {"PP", BCF_VL_G}, {"DP", BCF_VL_FIXED}, {"LEN", BCF_VL_FIXED}, {"FT", BCF_VL_FIXED}, {"GQ", BCF_VL_FIXED}, {"GT", BCF_VL_FIXED}, {"HQ", BCF_VL_FIXED},
    {"MQ", BCF_VL_FIXED}, {"PQ", BCF_VL_FIXED}, {"PS", BCF_VL_FIXED}, {nullptr, 0}};
const Spec kInfoSpec[] = {{"AD", BCF_VL_R}, {"ADF", BCF_VL_R}, {"ADR", BCF_VL_R}, {"AC", BCF_VL_A}, {"AF", BCF_VL_A}, {"CIGAR", BCF_VL_A}, {"AA", BCF_VL_FIXED},
    {"AN", BCF_VL_FIXED}, {"BQ", BCF_VL_FIXED}, {"DB", BCF_VL_FIXED}, {"DP", BCF_VL_FIXED}, {"END", BCF_VL_FIXED}, {"H2", BCF_VL_FIXED}, {"H3", BCF_VL_FIXED},
    {"MQ", BCF_VL_FIXED}, {"MQ0", BCF_VL_FIXED}, {"NS", BCF_VL_FIXED}, {"SB", BCF_VL_FIXED}, {"SOMATIC", BCF_VL_FIXED}, {"VALIDATED", BCF_VL_FIXED},
    {"1000G", BCF_VL_FIXED}, {nullptr, 0}};

bool list_after_correction(const Spec *tab, const std::string &name, int vl) {
    for (; tab->name; tab++) if (name == tab->name) {
        const bool wrong = tab->vl == BCF_VL_FIXED ? vl != BCF_VL_FIXED : (vl != tab->vl && vl != BCF_VL_VAR);
        if (wrong) vl = tab->vl;
        break;
    }
    return vl != BCF_VL_FIXED;
}

int duck_of(int ht) { return ht == BCF_HT_FLAG ? DT_BOOLEAN : ht == BCF_HT_INT ? DT_INTEGER : ht == BCF_HT_REAL ? DT_FLOAT : DT_VARCHAR; }

// bcftools split-vep type inference as the reference restates it (src/vep_parser.c:70-92)
int vep_infer_type(const std::string &n) {
    auto has = [&](const char *x) { return n.find(x) != std::string::npos; };
    if (n == "DISTANCE" || n == "STRAND" || n == "TSL" || n == "GENE_PHENO" || n == "HGVS_OFFSET" || n.compare(0, 9, "MOTIF_POS") == 0) return BCF_HT_INT;
    if (n == "Consequence" || n == "FLAGS" || n == "CLIN_SIG") return BCF_HT_STR;
    if (has("_AF") || has("AF_") || has("AFR_AF") || has("AMR_AF") || has("EAS_AF") || has("EUR_AF") || has("SAS_AF") || has("MAX_AF") || has("MOTIF_SCORE_CHANGE") ||
        n.compare(0, 17, "SpliceAI_pred_DS_") == 0) return BCF_HT_REAL;
    return BCF_HT_STR;
}

}  // namespace

int BcfHeader::find_id(const std::string &k) const {
    for (size_t i = 0; i < ids.size(); i++) if (ids[i].present && ids[i].key == k) return (int)i;
    return -1;
}

bool bcf_parse_header(const char *text, BcfHeader &h, std::string *err) {
    h = BcfHeader();
    auto failx = [&](const char *m) { if (err) *err = m; return false; };
    HeaderLine r; size_t len = 0;
    const char *p = text;
    // the first line is registered before the implicit PASS filter, then seen again by the main loop (vcf.c:1415-1429)
    if (parse_line(p, r, len) == 1) {
        if (!r.structured && r.key == "fileformat") h.version = version_of(r.value);
        if (register_line(h, r) < 0) return failx("conflicting IDX in header");
    }
    { HeaderLine pass; size_t l2; parse_line("##FILTER=<ID=PASS,Description=\"All filters passed\">", pass, l2); register_line(h, pass); }
    for (;;) {
        int rc = parse_line(p, r, len);
        if (rc < 0) return failx("unparsable header line");
        if (rc == 1) {
            if (!r.structured && !h.version && r.key == "fileformat") h.version = version_of(r.value);
            if (register_line(h, r) < 0) return failx("conflicting IDX in header");
            p += len; continue;
        }
        if (len > 0) { p += len; continue; }                  // malformed ## line: skipped
        if (!strncmp(p, "#CHROM\t", 7) || !strncmp(p, "#CHROM ", 7)) break;
        const char *eol = strchr(p, '\n');
        if (!eol) return failx("sample line not found");
        p = eol + 1;
    }
    if (!h.version) h.version = 4002000;
    static const char mand[] = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO";
    if (strncmp(p, mand, sizeof(mand) - 1)) return failx("bad #CHROM line");
    const char *b = p + sizeof(mand) - 1;
    if (*b && *b != '\n') {
        if (strncmp(b, "\tFORMAT\t", 8)) return failx("bad #CHROM line");
        b += 8;
        while (*b) {
            const char *e = b;
            while (*e && *e != '\t' && *e != '\n') e++;
            std::string nm(b, e);
            size_t ws = 0; while (ws < nm.size() && isspace((unsigned char)nm[ws])) ws++;
            if (ws == nm.size()) return failx("empty sample name");
            for (auto &s : h.samples) if (s == nm) return failx("duplicated sample name");
            h.samples.push_back(nm);
            if (!*e || *e == '\n') break;
            b = e + 1;
        }
    }
    static const char *vep[] = {"CSQ", "BCSQ", "ANN", "VEP", "vep"};
    for (const char *t : vep) { int id = h.find_id(t); if (id >= 0 && h.ids[id].has[BCF_HL_INFO]) h.has_vep_tag = true; }
    return true;
}

// one more "##" line into an existing header: what vcf_parse does for names a record uses without a definition (bcf_hdr_parse_line +
// bcf_hdr_add_hrec of a generated line, vcf.c:3744-3761, 3788-3803, 3846-3862); false when the line does not parse or register
bool bcf_header_add_line(BcfHeader &h, const char *line) {
    HeaderLine r; size_t len = 0;
    if (parse_line(line, r, len) != 1) return false;
    return register_line(h, r) >= 0;
}

void bcf_build_schema(const BcfHeader &h, bool tidy_format, BcfSchema &s) {
    s = BcfSchema();
    s.n_samples = (int)h.samples.size();
    s.tidy = tidy_format && s.n_samples > 0;
    s.gt_id = h.find_id("GT");
    s.gt_string_ok = s.gt_id >= 0 && h.ids[s.gt_id].has[BCF_HL_FMT] && h.ids[s.gt_id].type[BCF_HL_FMT] == BCF_HT_STR;
    auto add = [&](const std::string &n, int kind, int dt, bool list, int field = -1, int sample = -1) {
        BcfColumn c; c.name = n; c.kind = kind; c.duck_type = dt; c.is_list = list; c.field = field; c.sample = sample; s.cols.push_back(c);
    };
    add("CHROM", BK_CHROM, DT_VARCHAR, false); add("POS", BK_POS, DT_BIGINT, false); add("ID", BK_ID, DT_VARCHAR, false);
    add("REF", BK_REF, DT_VARCHAR, false); add("ALT", BK_ALT, DT_VARCHAR, true); add("QUAL", BK_QUAL, DT_DOUBLE, false);
    add("FILTER", BK_FILTER, DT_VARCHAR, true);
    // VEP / BCSQ / ANN: the first of CSQ, BCSQ, ANN, VEP, vep that is an INFO tag (vep_detect_tag, vep_parser.c:100-118); its Description
    // carries "Format: a|b|c" up to the closing quote (parse_format_string / split_format_fields :33-68); every field becomes a
    // VEP_<name> LIST column right behind FILTER (bcf_reader.c:582-603)
    int vep_id = -1;
    {
        static const char *tags[] = {"CSQ", "BCSQ", "ANN", "VEP", "vep"};
        for (const char *t : tags) { int id = h.find_id(t); if (id >= 0 && h.ids[id].has[BCF_HL_INFO]) { vep_id = id; s.vep_tag = t; break; } }
        if (vep_id >= 0) {
            const std::string &d = h.ids[vep_id].info_desc;
            size_t f = d.find("Format: ");
            int nf = 0;
            if (f != std::string::npos) {
                f += 8;
                size_t e = d.find('"', f); if (e == std::string::npos) e = d.size();
                nf = 1; for (size_t k = f; k < e; k++) if (d[k] == '|') nf++;
                if (nf <= 256) {
                    size_t st = f;
                    for (size_t k = f;; k++) {
                        if (k == e || d[k] == '|') { VepField vf; vf.name = d.substr(st, k - st); vf.htype = vep_infer_type(vf.name); s.vep_fields.push_back(vf); if (k == e) break; st = k + 1; }
                    }
                }
            }
            if (s.vep_fields.empty()) { vep_id = -1; s.vep_tag.clear(); }
            for (size_t v = 0; v < s.vep_fields.size(); v++) add(("VEP_" + s.vep_fields[v].name).substr(0, 255), BK_VEP, duck_of(s.vep_fields[v].htype), true, (int)v);
        }
    }
    for (size_t i = 0; i < h.ids.size(); i++) if (h.ids[i].present && h.ids[i].has[BCF_HL_INFO]) {
        if ((int)i == vep_id) s.vep_info_field = (int)s.info_fields.size();
        BcfField f; f.name = h.ids[i].key; f.id = (int)i; f.htype = h.ids[i].type[BCF_HL_INFO];
        f.is_list = list_after_correction(kInfoSpec, f.name, h.ids[i].vl[BCF_HL_INFO]);
        s.info_fields.push_back(f);
        add("INFO_" + f.name, BK_INFO, duck_of(f.htype), f.is_list, (int)s.info_fields.size() - 1);
    }
    if (s.n_samples > 0) {
        for (size_t i = 0; i < h.ids.size(); i++) if (h.ids[i].present && h.ids[i].has[BCF_HL_FMT]) {
            BcfField f; f.name = h.ids[i].key; f.id = (int)i; f.htype = h.ids[i].type[BCF_HL_FMT];
            f.is_list = list_after_correction(kFmtSpec, f.name, h.ids[i].vl[BCF_HL_FMT]);
            s.format_fields.push_back(f);
        }
        if (s.format_fields.empty()) { BcfField f; f.name = "GT"; f.id = -1; f.htype = BCF_HT_STR; f.is_list = false; s.format_fields.push_back(f); }   // bcf_reader.c:683-692
        if (tidy_format) {
            add("SAMPLE_ID", BK_SAMPLE_ID, DT_VARCHAR, false);
            for (size_t f = 0; f < s.format_fields.size(); f++)
                add("FORMAT_" + s.format_fields[f].name, BK_FORMAT, duck_of(s.format_fields[f].htype), s.format_fields[f].is_list, (int)f, -1);
        } else {
            for (int sm = 0; sm < s.n_samples; sm++) for (size_t f = 0; f < s.format_fields.size(); f++)
                add("FORMAT_" + s.format_fields[f].name + "_" + h.samples[sm], BK_FORMAT, duck_of(s.format_fields[f].htype), s.format_fields[f].is_list, (int)f, sm);
        }
    }
}

}  // namespace dhts
