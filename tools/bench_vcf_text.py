#!/usr/bin/env python3
"""read_bcf on bgzipped VCF TEXT through the table function: the shape of the reference's published read_bcf numbers (Benchmark.md:517-523:
ClinVar vcf.gz, 189 MB compressed, 4.35 M sites-only records with a dozen INFO keys; COUNT(*) 1.234 s, CHROM/POS/REF/ALT and six INFO
columns with LIMIT 200 000 in 0.26-0.29 s, unstated x86-64, 4 DuckDB threads).  A synthetic file of that shape is generated here."""
import json
import os
import random
import subprocess
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import duckhts_amd  # noqa: E402
import bamwriter  # noqa: E402

HOST = os.path.join(ROOT, "tests", "minihost", "minihost")
HDR = ["##fileformat=VCFv4.1", "##fileDate=2025-01-01", "##source=ClinVar", "##reference=GRCh38"] + \
      ['##INFO=<ID=%s,Number=%s,Type=%s,Description="x">' % kv for kv in (("AF_ESP", "1", "Float"), ("AF_EXAC", "1", "Float"), ("AF_TGP", "1", "Float"), ("ALLELEID", "1", "Integer"),
       ("CLNDN", ".", "String"), ("CLNDISDB", ".", "String"), ("CLNHGVS", ".", "String"), ("CLNREVSTAT", ".", "String"), ("CLNSIG", ".", "String"), ("CLNVC", "1", "String"),
       ("CLNVCSO", "1", "String"), ("GENEINFO", "1", "String"), ("MC", ".", "String"), ("ORIGIN", ".", "String"), ("RS", ".", "String"))] + \
      ["##contig=<ID=%s>" % c for c in list(map(str, range(1, 23))) + ["X", "Y", "MT"]] + ["#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"]


def generate(path, n, seed=1):
    rnd = random.Random(seed)
    chroms = list(map(str, range(1, 23))) + ["X", "Y", "MT"]
    per = n // len(chroms) + 1
    co = []
    buf = ["\n".join(HDR) + "\n"]
    size = 0
    with open(path, "wb") as f:
        def flush(final=False):
            nonlocal buf
            raw = "".join(buf).encode()
            buf = []
            f.write(bamwriter.bgzf_file(raw, eof=final, level=6))
        k = 0
        for c in chroms:
            pos = 10000
            for _ in range(per):
                if k >= n:
                    break
                pos += rnd.randrange(1, 700)
                ref = rnd.choice("ACGT"); alt = rnd.choice([x for x in "ACGT" if x != ref])
                aid = 15000 + k
                info = [f"ALLELEID={aid}", "CLNDISDB=MedGen:C%07d,OMIM:%06d" % (rnd.randrange(10 ** 7), rnd.randrange(10 ** 6)), "CLNDN=" + rnd.choice(["not_provided", "Hereditary_cancer-predisposing_syndrome", "Inborn_genetic_diseases", "not_specified"]),
                        f"CLNHGVS=NC_0000{len(c):02d}.11:g.{pos}{ref}>{alt}", "CLNREVSTAT=criteria_provided,_single_submitter", "CLNSIG=" + rnd.choice(["Benign", "Likely_benign", "Uncertain_significance", "Pathogenic"]),
                        "CLNVC=single_nucleotide_variant", "CLNVCSO=SO:0001483", f"GENEINFO=GENE{rnd.randrange(20000)}:{rnd.randrange(100000)}", "MC=SO:0001583|missense_variant", "ORIGIN=1"]
                if rnd.random() < 0.3:
                    info.insert(0, "AF_EXAC=%.5f" % rnd.random())
                if rnd.random() < 0.5:
                    info.append("RS=%d" % rnd.randrange(10 ** 9))
                buf.append(f"{c}\t{pos}\t{aid}\t{ref}\t{alt}\t.\t.\t{';'.join(info)}\n")
                k += 1
                if len(buf) >= 200000:
                    flush()
        flush(final=True)
    return k


def generate_gnomad_shape(path, n, seed=2):
    """the shape of Benchmark.md:801 (gnomAD exomes chr22 with VEP: 1.4 GB of vcf.bgz for 416,083 records, i.e. ~3.4 KB compressed and tens
    of KB of text per line: hundreds of numeric INFO keys and a long `vep` string per record)"""
    rnd = random.Random(seed)
    pops = ["afr", "amr", "asj", "eas", "fin", "nfe", "oth", "sas"]
    keys = [("AC", "A", "Integer"), ("AN", "1", "Integer"), ("AF", "A", "Float")]
    for sub in ["", "non_neuro_", "non_cancer_", "controls_", "non_topmed_"]:
        for pop in pops:
            for sex in ["", "_female", "_male"]:
                for k, t in (("AC", "Integer"), ("AN", "Integer"), ("AF", "Float"), ("nhomalt", "Integer")):
                    keys.append((f"{sub}{k}_{pop}{sex}", "A" if k != "AN" else "1", t))
    keys += [("FS", "1", "Float"), ("MQ", "1", "Float"), ("QD", "1", "Float"), ("VQSLOD", "1", "Float"), ("ReadPosRankSum", "1", "Float"), ("DP", "1", "Integer")]
    vep_fmt = "Allele|Consequence|IMPACT|SYMBOL|Gene|Feature_type|Feature|BIOTYPE|EXON|INTRON|HGVSc|HGVSp|cDNA_position|CDS_position|Protein_position|Amino_acids|Codons|Existing_variation|ALLELE_NUM|DISTANCE|STRAND|FLAGS|VARIANT_CLASS|SYMBOL_SOURCE|HGNC_ID|CANONICAL|TSL|APPRIS|CCDS|ENSP|SWISSPROT|TREMBL|UNIPARC|GENE_PHENO|SIFT|PolyPhen|DOMAINS|HGVS_OFFSET|LoF|LoF_filter|LoF_flags|LoF_info"
    hdr = ["##fileformat=VCFv4.2", '##FILTER=<ID=AC0,Description="x">', '##FILTER=<ID=RF,Description="x">'] + \
          ['##INFO=<ID=%s,Number=%s,Type=%s,Description="x">' % k for k in keys] + \
          ['##INFO=<ID=vep,Number=.,Type=String,Description="Consequence annotations from Ensembl VEP. Format: %s">' % vep_fmt, "##contig=<ID=22,length=50818468>", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"]
    nv = vep_fmt.count("|") + 1
    with open(path, "wb") as f:
        buf = ["\n".join(hdr) + "\n"]; pos = 16000000
        for k in range(n):
            pos += rnd.randrange(1, 200)
            ref = rnd.choice("ACGT"); alt = rnd.choice([x for x in "ACGT" if x != ref])
            info = []
            for name, num, typ in keys:
                info.append(f"{name}={rnd.randrange(250000)}" if typ == "Integer" else f"{name}=%.5e" % rnd.random())
            trs = []
            for _ in range(rnd.randrange(3, 40)):
                t = [alt, rnd.choice(["missense_variant", "intron_variant", "downstream_gene_variant"]), "MODIFIER", "GENE%d" % rnd.randrange(900), "ENSG%011d" % rnd.randrange(10 ** 9), "Transcript",
                     "ENST%011d" % rnd.randrange(10 ** 9), "protein_coding"] + [rnd.choice(["", "", "x%d" % rnd.randrange(1000)]) for _ in range(nv - 8)]
                trs.append("|".join(t))
            info.append("vep=" + ",".join(trs))
            buf.append(f"22\t{pos}\trs{k}\t{ref}\t{alt}\t{rnd.randrange(100000)}.{rnd.randrange(100)}\t{rnd.choice(['PASS', 'AC0', 'RF'])}\t{';'.join(info)}\n")
            if len(buf) >= 4000:
                f.write(bamwriter.bgzf_file("".join(buf).encode(), eof=False, level=6)); buf = []
        f.write(bamwriter.bgzf_file("".join(buf).encode(), eof=True, level=6))
    return n


def run(path, proj, threads, repeat=4, env=None):
    cmd = [HOST, duckhts_amd.LIB_PATH, "read_bcf", path, "-t", str(threads), "-r", str(repeat), "-p", ",".join(map(str, proj))]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout + r.stderr
    rows = int(r.stdout.split("OK rows=")[1].split()[0])
    return rows, [float(l.split("seconds=")[1].split()[0]) for l in r.stdout.splitlines() if l.startswith("RUN ")]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_352_930
    d = tempfile.mkdtemp(dir="/tmp")
    path = os.path.join(d, "clinvar_like.vcf.gz")
    t0 = time.time()
    generate(path, n)
    size = os.path.getsize(path)
    print(json.dumps({"generated": path, "records": n, "compressed_bytes": size, "seconds": round(time.time() - t0, 1)}), flush=True)
    names = ["CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER"] + ["INFO_" + x for x in ("AF_ESP", "AF_EXAC", "AF_TGP", "ALLELEID", "CLNDN", "CLNDISDB", "CLNHGVS", "CLNREVSTAT", "CLNSIG", "CLNVC", "CLNVCSO", "GENEINFO", "MC", "ORIGIN", "RS")]
    queries = [("COUNT(*) (Benchmark.md:517: 1.234 s)", [0]), ("CHROM,POS,REF,ALT (Benchmark.md:518: LIMIT 200000 in 0.257 s)", [0, 1, 3, 4]),
               ("6 INFO columns (Benchmark.md:519: LIMIT 200000 in 0.294 s)", [names.index("INFO_" + x) for x in ("ALLELEID", "CLNSIG", "CLNVC", "GENEINFO", "MC", "RS")]),
               ("all 22 columns", list(range(len(names))))]
    for qn, proj in queries:
        for thr in (1, 4):
            for cache in ("0", "1"):
                rows, runs = run(path, proj, thr, env={"DHTS_THREADS": str(thr), "DHTS_FILE_CACHE": cache})
                warm = sorted(runs[1:])[len(runs[1:]) // 2]
                print(json.dumps({"operator": "read_bcf on bgzipped VCF text through the DuckDB table function (mini host), full scan", "query": qn, "rows": rows, "DHTS_THREADS": thr,
                                  "file_resident": cache == "1", "first_query_s": round(runs[0], 3), "warm_query_s": round(warm, 4), "rows_per_s": round(rows / warm, 1),
                                  "compressed_MBps": round(size / warm / 1e6, 1)}), flush=True)


def main_region():
    """region queries on the ClinVar-shaped file: tabix index written by the library, then read_bcf(region := ...) through the table function
    with the default (header blocks + index windows staged) and with DHTS_SPARSE=0 (whole file staged), file read every query"""
    import ctypes as C
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_352_930
    d = tempfile.mkdtemp(dir="/tmp")
    path = os.path.join(d, "clinvar_like.vcf.gz")
    generate(path, n)
    size = os.path.getsize(path)
    L = duckhts_amd.lib()
    L.dhts_bcf_build_index.restype = C.c_int64; L.dhts_bcf_build_index.argtypes = [C.c_void_p, C.c_int]
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(path); ctx.bgzf_index(); duckhts_amd.BcfScan(ctx)
        t0 = time.time(); k = L.dhts_bcf_build_index(ctx.h, 0); dt = time.time() - t0
        assert k > 0, L.dhts_error(ctx.h)
        import numpy as np
        raw = np.zeros(k, np.uint8); L.dhts_bam_index_bytes(ctx.h, raw.ctypes.data, k)
        open(path + ".tbi", "wb").write(ctx.bgzf_compress(raw.tobytes()))
    finally:
        ctx.close()
    print(json.dumps({"generated": path, "records": n, "compressed_bytes": size, "tbi_bytes": os.path.getsize(path + ".tbi"), "tbi_build_seconds": round(dt, 3)}), flush=True)
    for rg in ("1:1000000-1100000", "7:50000000-60000000", "X", "1:1-250000000,2:1-1000,22:30000000-31000000"):
        for sparse in ("1", "0"):
            cmd = [HOST, duckhts_amd.LIB_PATH, "read_bcf", path, "-t", "1", "-r", "5", "-p", "0", "-n", "region=" + rg]
            r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, DHTS_FILE_CACHE="0", DHTS_SPARSE=sparse))
            assert r.returncode == 0, r.stdout + r.stderr
            rows = int(r.stdout.split("OK rows=")[1].split()[0])
            runs = [float(l.split("seconds=")[1].split()[0]) for l in r.stdout.splitlines() if l.startswith("RUN ")]
            warm = sorted(runs[1:])[len(runs[1:]) // 2]
            print(json.dumps({"operator": "read_bcf(region) on bgzipped VCF text through the table function, COUNT(*), file read every query", "region": rg, "rows": rows,
                              "staging": "header + index windows" if sparse == "1" else "whole file (DHTS_SPARSE=0)", "warm_query_s": round(warm, 4), "first_query_s": round(runs[0], 3)}), flush=True)


def generate_samples_shape(path, n, n_smp, seed=3):
    """a cohort call set: GT:DP:GQ:AD:PL for n_smp samples per line (1000 Genomes / gVCF-merge shape: the line's length is the samples')"""
    rnd = random.Random(seed)
    hdr = ["##fileformat=VCFv4.2", '##FILTER=<ID=LowQual,Description="x">', '##INFO=<ID=AC,Number=A,Type=Integer,Description="x">', '##INFO=<ID=AF,Number=A,Type=Float,Description="x">',
           '##INFO=<ID=AN,Number=1,Type=Integer,Description="x">', '##INFO=<ID=DP,Number=1,Type=Integer,Description="x">',
           '##FORMAT=<ID=GT,Number=1,Type=String,Description="x">', '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="x">', '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="x">',
           '##FORMAT=<ID=AD,Number=R,Type=Integer,Description="x">', '##FORMAT=<ID=PL,Number=G,Type=Integer,Description="x">', "##contig=<ID=20,length=64444167>",
           "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("S%05d" % i for i in range(n_smp))]
    pool = []
    for _ in range(400):
        gt = rnd.choice(["0|0", "0|0", "0|0", "0|1", "1|0", "1|1", "./."])
        if gt == "./.":
            pool.append("./.:.:.:.:.")
        else:
            a, b = rnd.randrange(40), rnd.randrange(30)
            pool.append("%s:%d:%d:%d,%d:%d,%d,%d" % (gt, a + b, rnd.randrange(100), a, b, rnd.randrange(300), rnd.randrange(300), rnd.randrange(300)))
    with open(path, "wb") as f:
        buf = ["\n".join(hdr) + "\n"]; pos = 60000; nbuf = 0
        for k in range(n):
            pos += rnd.randrange(1, 300)
            ref = rnd.choice("ACGT"); alt = rnd.choice([x for x in "ACGT" if x != ref])
            line = "20\t%d\t.\t%s\t%s\t%d\tPASS\tAC=%d;AF=%.4f;AN=%d;DP=%d\tGT:DP:GQ:AD:PL\t%s\n" % (pos, ref, alt, rnd.randrange(9000), rnd.randrange(2 * n_smp), rnd.random(), 2 * n_smp,
                                                                                                   rnd.randrange(50 * n_smp), "\t".join(rnd.choices(pool, k=n_smp)))
            buf.append(line); nbuf += len(line)
            if nbuf >= 32 << 20:
                f.write(bamwriter.bgzf_file("".join(buf).encode(), eof=False, level=6)); buf = []; nbuf = 0
        f.write(bamwriter.bgzf_file("".join(buf).encode(), eof=True, level=6))
    return n


def main_samples():
    """the scan through the library (tidy_format: a row per record and sample), columns left in HBM; the mini host stops at 1024 result columns"""
    n_smp = int(sys.argv[2]) if len(sys.argv) > 2 else 2504
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
    d = tempfile.mkdtemp(dir="/tmp")
    path = os.path.join(d, "cohort_like.vcf.gz")
    t0 = time.time()
    generate_samples_shape(path, n, n_smp)
    size = os.path.getsize(path)
    print(json.dumps({"generated": path, "records": n, "samples": n_smp, "compressed_bytes": size, "seconds": round(time.time() - t0, 1)}), flush=True)
    for tidy, qn, proj in ((False, "COUNT(*) (wide)", ["CHROM"]), (False, "CHROM,POS,REF,ALT,INFO_AF (wide)", ["CHROM", "POS", "REF", "ALT", "INFO_AF"]),
                           (True, "tidy: CHROM,POS,SAMPLE_ID,FORMAT_GT", ["CHROM", "POS", "SAMPLE_ID", "FORMAT_GT"]),
                           (True, "tidy: CHROM,POS,SAMPLE_ID,FORMAT_GT,FORMAT_DP,FORMAT_GQ,FORMAT_AD,FORMAT_PL", ["CHROM", "POS", "SAMPLE_ID", "FORMAT_GT", "FORMAT_DP", "FORMAT_GQ", "FORMAT_AD", "FORMAT_PL"])):
        ctx = duckhts_amd.Context(0)
        try:
            ctx.open(path); ctx.bgzf_index()
            sc = duckhts_amd.BcfScan(ctx, tidy=tidy)
            sc.set_projection(proj)
            times = []
            for rep in range(4):
                ctx.L.dhts_sync(ctx.h); t0 = time.perf_counter()
                if rep:
                    sc.rewind()
                rows = 0
                while True:
                    b = sc.next_batch(0)
                    rows += b.n_rows
                    if b.status != 0:
                        assert b.status > 0, b.status
                        break
                ctx.L.dhts_sync(ctx.h); times.append(time.perf_counter() - t0)
            warm = sorted(times[1:])[1]
            assert rows == n * (n_smp if tidy else 1), rows
            print(json.dumps({"scan": "read_bcf on a cohort-shaped vcf.gz (%d samples), file resident, columns left in HBM" % n_smp, "query": qn, "rows": rows, "first_s": round(times[0], 3),
                              "warm_s": round(warm, 4), "records_per_s": round(n / warm, 1), "genotypes_per_s": round(n * n_smp / warm, 1), "compressed_MBps": round(size / warm / 1e6, 1)}), flush=True)
        finally:
            ctx.close()


def main_gnomad():
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    d = tempfile.mkdtemp(dir="/tmp")
    path = os.path.join(d, "gnomad_like.vcf.bgz")
    t0 = time.time()
    generate_gnomad_shape(path, n)
    size = os.path.getsize(path)
    print(json.dumps({"generated": path, "records": n, "compressed_bytes": size, "compressed_bytes_per_record": round(size / n, 1), "seconds": round(time.time() - t0, 1)}), flush=True)
    for qn, proj in (("COUNT(*) (Benchmark.md:801: 29.91 s for 416,083 records = 13.9 k rows/s, 44.8 MB/s)", [0]), ("CHROM,POS,REF,ALT,VEP_SYMBOL,VEP_Consequence", [0, 1, 3, 4, 10, 8])):
        for thr in (1, 4):
            for cache in ("0", "1"):                          # file read and copied every query / still resident in HBM from the previous query
                rows, runs = run(path, proj, thr, env={"DHTS_THREADS": str(thr), "DHTS_FILE_CACHE": cache})
                warm = sorted(runs[1:])[len(runs[1:]) // 2]
                print(json.dumps({"operator": "read_bcf on a gnomAD-shaped vcf.bgz through the DuckDB table function (mini host), full scan", "query": qn, "rows": rows, "DHTS_THREADS": thr,
                                  "file": "read every query" if cache == "0" else "resident in HBM", "first_query_s": round(runs[0], 3), "warm_query_s": round(warm, 4),
                                  "rows_per_s": round(rows / warm, 1), "compressed_MBps": round(size / warm / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "gnomad":
        main_gnomad()
    elif len(sys.argv) > 1 and sys.argv[1] == "samples":
        main_samples()
    elif len(sys.argv) > 1 and sys.argv[1] == "region":
        main_region()
    else:
        main()
