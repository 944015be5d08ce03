"""The two-bit packer of FastK_amd's reader threads (scan_text_packed in fastk_amd/csrc/host/FastK_amd.c: the bases of
a FASTA / FASTQ piece four to a byte, stretches of non-bases listed beside them -- what fk_push_packed takes) against a
base-by-base restatement: tests/csrc/host_packer_check.c includes the driver's source and drives pk_bases on random
reads at every bit offset, through the AVX2 path and the table path; and the piece parser (pk_parse_piece) against the
reference's scanner (io.c:685-738) restated one byte at a time, on random FASTA / FASTQ texts with records without bases,
empty lines, '>' and '@' where they mislead, a missing final newline and truncated ends.  No GPU needed."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reader_thread_packer_matches_restatement(tmp_path):
    lib = os.path.join(ROOT, "fastk_amd", "lib")
    assert os.path.exists(os.path.join(lib, "libfastk_amd.so")), "build fastk_amd/csrc first"
    exe = str(tmp_path / "host_packer_check")
    subprocess.run(["gcc", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "csrc", "host_packer_check.c"),
                    os.path.join(ROOT, "fastk_amd", "csrc", "host", "input_formats.c"),      # (the driver's SAM / BAM readers)
                    "-L" + lib, "-lfastk_amd", "-lz", "-lpthread", "-Wl,-rpath," + lib], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "packer OK" in out.stdout, out.stdout + out.stderr
