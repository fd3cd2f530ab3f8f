"""Metagenome inputs: genome list (-gl), abundance table (-a), DNA type list (-dl).

Mirrors the metagenome half of ``read_profile`` (src/simulator.py:257-266, 284-339, 357-380) and the abundance helpers
``add_abundance_var`` (S:594-615) / ``inflate_abun`` (S:2018-2022).  Chromosomes of all species are concatenated into one
:class:`~nanosim_amd.model.Reference`; the chromosome name handed to the engine is ``"<species>-<chrom>"`` — what
``extract_read`` puts in the read name (S:1747).
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass

import numpy as np

from .model import Reference, normalise_name


@dataclass
class MetaReference:
    ref: Reference
    species: list[str]
    species_chrom_off: np.ndarray          # uint32 [nspecies+1]: chromosomes of species s are [off[s], off[s+1])
    chrom_names: list[list[str]]           # per species, normalised chromosome keys (dict order of the reference)

    @property
    def max_chrom(self) -> int:
        return self.ref.max_chrom          # max over species of max_chrom[species] (S:2524)

    def total_len(self) -> dict:
        out = {}
        lens = np.diff(self.ref.chrom_off.astype(np.int64))
        for i, sp in enumerate(self.species):
            out[sp] = int(lens[self.species_chrom_off[i]:self.species_chrom_off[i + 1]].sum())
        return out


def species_key(name: str) -> str:
    return "_".join(name.split())                                           # S:263, 333, 374


def read_genome_list(path: str) -> dict:
    ref = {}
    with open(path) as f:                                                   # S:259-266
        for line in f.readlines():
            fields = line.split("\t")
            if len(fields) < 2:
                continue
            ref[species_key(fields[0])] = fields[1].strip("\n")
    return ref


def _read_fasta_records(path: str):
    """(normalised name, uint8 bases) per record, readfq semantics for FASTA (S:709-740, 320-323)."""
    from .model import read_fasta
    r = read_fasta(path)
    return [(nm, r.chrom(i)) for i, nm in enumerate(r.names)]


def read_metagenome(genome_list: str, dna_type_list: str | None) -> MetaReference:
    ref = read_genome_list(genome_list)
    base_dirs = [os.getcwd(), os.path.dirname(os.path.abspath(genome_list))]
    species, names, chunks, circ, off, keys = [], [], [], [], [0], []
    dna_type = {}
    for sp, fq_path in ref.items():
        if fq_path.startswith(("ftp", "http")):
            raise ValueError("streaming references from RefSeq (S:295-315) is not supported (no network in this build)")
        path = fq_path
        if not os.path.isabs(path) and not os.path.exists(path):
            for b in base_dirs:
                if os.path.exists(os.path.join(b, fq_path)):
                    path = os.path.join(b, fq_path)
                    break
        recs = _read_fasta_records(path)
        species.append(sp)
        keys.append([k for k, _ in recs])
        dna_type[sp] = {k: "circular" for k, _ in recs}                    # circular as default (S:325)
        for k, seq in recs:
            names.append(sp + "-" + k)
            chunks.append(seq)
        off.append(off[-1] + len(recs))
    if dna_type_list:                                                       # S:328-339
        with open(dna_type_list) as f:
            for line in f.readlines():
                fields = line.split("\t")
                if len(fields) < 3:
                    continue
                sp = species_key(fields[0])
                chr_name = "-".join(re.split(r"[_\s]\s*", fields[1].partition(" ")[0])).split(".")[0]
                if sp not in ref:
                    raise SystemExit("You didn't provide a reference genome for " + sp)
                dna_type[sp][chr_name] = fields[2].strip("\n")
    for sp, ks in zip(species, keys):
        for k in ks:
            circ.append(dna_type[sp][k] == "circular")
    lens = np.array([len(c) for c in chunks], dtype=np.uint64)
    r = Reference(names, np.ascontiguousarray(np.concatenate(chunks).astype(np.uint8)),
                  np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64), np.array(circ, dtype=np.uint8))
    return MetaReference(r, species, np.array(off, dtype=np.uint32), keys)


def read_abundance(path: str, species: list[str]):
    """-> (reads per sample, [ {species: abundance} per sample ])   (S:357-380)"""
    with open(path) as f:
        header = f.readline()
        number_list = [int(x) for x in header.strip().split("\t")[1:]]
        n = len(number_list)
        samples = [dict() for _ in range(n)]
        for line in f.readlines():
            fields = line.split("\t")
            if not line.strip():
                continue
            if n != len(fields) - 1:
                raise SystemExit("Abundance file is incorrectly formatted. Check that each row has the same number of columns")
            sp = species_key(fields[0])
            if sp not in species:
                raise SystemExit("You didn't provide a reference genome for " + sp)
            for i in range(n):
                samples[i][sp] = float(fields[1 + i])
    return number_list, samples


def add_abundance_var(expected: dict, total_len: dict, var_low: float, var_high: float, uniforms) -> dict:
    """S:594-615; `uniforms` yields the U(0,1) draws behind random.uniform(var_low, var_high)."""
    abun_var = [var_low + (var_high - var_low) * next(uniforms) for _ in range(len(total_len))]
    per_species = {}
    for var, sp in zip(sorted(abun_var, key=abs), sorted(total_len, key=lambda k: total_len[k])):
        per_species[sp] = var
    out = {sp: e + e * per_species[sp] for sp, e in expected.items()}
    total = sum(out.values())
    return {sp: a * 100 / total for sp, a in out.items()}


def inflate_abun(abun: dict, species: str, abun_inflation: float) -> float:
    return 1 - (1 - abun[species]) * abun_inflation                        # S:2018-2022
