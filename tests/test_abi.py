"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/raygraph.h declares, refuses to work without a GPU, and its host-side helpers agree with the
oracle.  No compute entry point is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "raygraph.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rg_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ray_amd import _lib
    L = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(L, name), "libraygraph.so does not export %s" % name
    assert sorted(_lib.SYMBOLS) == names, "ray_amd/_lib.py binds a different set than the header declares"
    assert b"gfx950" in L.rg_version()


def test_header_cites_the_reference_for_every_entry_point():
    text = open(os.path.join(ROOT, "include", "raygraph.h")).read()
    for cite in ("KmerAcademyBuilder.cpp:43-211", "MessageProcessor.cpp:692-809", "VerticesExtractor.cpp:40-274",
                 "CoverageGatherer.cpp:124-190", "GridTable.cpp:84-109", "Vertex.cpp:438-454", "EdgePurgerWorker.cpp:28-102",
                 "CoverageDistribution.cpp:33-112", "Read.cpp:31-159"):
        assert cite in text


def test_no_cpu_fallback():
    """without a HIP device rg_create must fail loudly; with one this is covered by the gpu tests"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from ray_amd import GraphBuilder, RayGraphError
    with pytest.raises(RayGraphError) as e:
        GraphBuilder(21)
    assert e.value.code == -2 and "no CPU path" in str(e.value)


def test_product_never_touches_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ray_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "ray_oracle" not in text and "oracle/" not in text, f


def test_bad_configurations_are_rejected():
    from ray_amd import _lib
    L = _lib.load()
    for k, words in ((22, 0), (0, 0), (33, 1), (65, 2), (21, 3)):
        cfg = _lib.RgConfig(k=k, kmer_words=words, rank=0, size=1, device=0, bloom_bits=0, table_slots=0,
                            first_read_id=0, stream=None)
        h = C.c_void_p()
        assert L.rg_create(C.byref(cfg), C.byref(h)) == _lib.RG_ERR_ARG
        assert L.rg_last_error(None) != b""
    cfg = _lib.RgConfig(k=21, kmer_words=0, rank=2, size=2, device=0, bloom_bits=0)
    assert L.rg_create(C.byref(cfg), C.byref(h)) == _lib.RG_ERR_ARG


def test_pack_read_matches_oracle(oracle):
    import ray_amd
    rng = np.random.default_rng(3)
    alphabet = "ACGTacgtNnxR"
    cases = ["", "A", "N", "NNNN", "ACGT", "NNACGTNN", "acgtn", "ACNNGT", "xACGTx", "TTTTT"]
    for _ in range(200):
        cases.append("".join(alphabet[i] for i in rng.integers(0, len(alphabet), rng.integers(0, 90))))
    for s in cases:
        for trim in (True, False):
            assert ray_amd.pack_read(s, trim) == oracle.read_pack(s, trim), (s, trim)


def test_coverage_distribution_matches_oracle(oracle, tmp_path):
    import ray_amd
    rng = np.random.default_rng(11)
    for trial in range(60):
        n = int(rng.integers(1, 120))
        cov = np.sort(rng.choice(np.arange(1, 400 if trial % 3 else 70000), size=n, replace=False)).astype(np.uint32)
        cnt = (rng.integers(1, 80000, size=n) * 2).astype(np.uint64)
        if trial % 4 == 0:
            cnt[rng.integers(0, n)] = 200000
        a, b = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
        assert ray_amd.coverage_distribution(cov, cnt, a) == oracle.coverage_distribution(cov, cnt, b)
        assert open(a, "rb").read() == open(b, "rb").read()
    with pytest.raises(ray_amd.RayGraphError):
        ray_amd.coverage_distribution(np.zeros(0, np.uint32), np.zeros(0, np.uint64))


def test_header_is_plain_c_and_the_glue_calls_type_check(tmp_path):
    """include/raygraph.h must compile as C (the boundary has no C++ types), and the call sequence INTEGRATION.md shows
    -- one rg_handle* member, rg_create(&cfg, &member), every later call with the member itself -- must type-check."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    src = tmp_path / "glue.c"
    src.write_text("""
#include <raygraph.h>
#include <string.h>
static rg_handle *m_gpuGraph;
static uint64_t rayHash1(const uint64_t *w, void *user){ (void)user; return w[0]; }   /* stands for Kmer::hash_function_1 */
static uint64_t rayHash2(const uint64_t *w, void *user){ (void)user; return w[0]; }
int glue(const uint8_t *packed, uint64_t nbytes, const uint64_t *off, const uint32_t *len, uint64_t n, void *d_buf){
	rg_config cfg; memset(&cfg,0,sizeof cfg); cfg.k=31; cfg.size=1; cfg.bloom_bits=RG_BLOOM_DEFAULT;
	if(rg_create(&cfg,&m_gpuGraph)!=RG_OK) return 1;
	if(rg_check_hashes(m_gpuGraph,rayHash1,rayHash2,NULL,4096)==RG_ERR_HASH) return 3;
	rg_add_reads(m_gpuGraph,packed,nbytes,off,len,n);
	if(rg_count_kmers(m_gpuGraph)!=RG_OK) return 2;
	uint64_t nh=0; rg_histogram(m_gpuGraph,NULL,NULL,0,&nh);
	rg_add_edges(m_gpuGraph);
	uint64_t pairs=0; rg_export(m_gpuGraph,NULL,0,&pairs);
	uint64_t counts[1]; uint32_t rb; rg_records_local_part(m_gpuGraph,0,4); rg_record_counts(m_gpuGraph,counts,&rb);
	rg_export_records(m_gpuGraph,d_buf,counts[0]); rg_count_records(m_gpuGraph,d_buf,counts[0]);
	rg_owner_counts(m_gpuGraph,counts); rg_resolve_admission(m_gpuGraph,n);
	rg_markers_query_counts(m_gpuGraph,counts); rg_markers_export_queries(m_gpuGraph,d_buf,counts[0]);
	rg_markers_answer(m_gpuGraph,d_buf,counts[0],(uint32_t*)d_buf); rg_markers_apply(m_gpuGraph,(const uint32_t*)d_buf,counts[0],(int32_t*)d_buf,(int32_t*)d_buf);
	return rg_destroy(m_gpuGraph);
}
/* size > 1: the communicator and the exchanges, as INTEGRATION.md's glue calls them */
static rg_comm *m_gpuComm;
static int a2a(void *u,const uint64_t *s,uint64_t *r,uint32_t n){ (void)u;(void)s;(void)r;(void)n; return 0; }
static int a2av(void *u,const void *ds,const uint64_t *sc,void *dr,const uint64_t *rc,uint32_t unit){ (void)u;(void)ds;(void)sc;(void)dr;(void)rc;(void)unit; return 0; }
int glue_ranks(int rank,int size,int device,uint64_t myReads,int32_t *d_fwd,int32_t *d_rev){
	unsigned char id[RG_COMM_ID_BYTES];
	uint64_t nh=0, sizes[64];
	rg_transport t; rg_comm_stats cs;
	if(rank==0 && rg_comm_unique_id(id)!=RG_OK) return 1;
	if(rg_comm_create_rccl(id,rank,size,device,&m_gpuComm)!=RG_OK){ (void)rg_comm_last_error(NULL); return 2; }
	t.user=NULL; t.alltoall_u64=a2a; t.alltoallv=a2av;
	if(rg_comm_create_transport(&t,rank,size,device,&m_gpuComm)!=RG_OK) return 3;
	if(rg_count_kmers_exchange(m_gpuGraph,m_gpuComm,size==2?RG_ROUTE_ENTRIES:RG_ROUTE_RECORDS,4,myReads)==RG_ERR_COMM) return 4;
	rg_exchange_records(m_gpuGraph,m_gpuComm,4); rg_exchange_owner_buckets(m_gpuGraph,m_gpuComm); rg_resolve_admission(m_gpuGraph,myReads);
	rg_histogram_exchange(m_gpuGraph,m_gpuComm,NULL,NULL,0,&nh);
	rg_purge_edges_exchange(m_gpuGraph,m_gpuComm);
	rg_read_markers_exchange(m_gpuGraph,m_gpuComm,4,d_fwd,d_rev);
	rg_allgather_u64(m_gpuGraph,m_gpuComm,nh,sizes);
	rg_comm_get_stats(m_gpuComm,&cs);
	return rg_comm_destroy(m_gpuComm);
}
""")
    for compiler, std in (("gcc", "-std=c99"), ("g++", "-std=c++98")):
        res = subprocess.run([compiler, std, "-Wall", "-Werror", "-fsyntax-only", "-x", "c" if compiler == "gcc" else "c++", "-I", inc, str(src)],
                             capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
