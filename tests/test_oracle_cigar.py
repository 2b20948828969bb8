"""CPU oracle (oracle/) against the golden vectors produced by the reference itself.  No GPU needed."""
import pytest

import util
from oracle import oracle


def test_homology_known_answers(built):
    for k in util.kat()['homology']:
        f = oracle.left_homology if k['dir'] == 'L' else oracle.right_homology
        assert f(k['pos'], k['seq'], k['sv']) == k['value'], k


def test_tokenizer_known_answers(built):
    for k in util.kat()['tokenize']:
        rc, tuples, _, _ = oracle.cigar_tokenize(k['cigar'])
        assert rc == 0
        assert [[l, o] for l, o in tuples] == k['tuples']


@pytest.mark.parametrize('case', ['cigar_synth', 'cigar_edge'])
def test_tables_byte_exact(built, case):
    d, df_align, df_trim = util.golden_case(case)
    df_snv, df_insdel = util.oracle_frames(d, df_align, df_trim)
    assert util.frame_text(df_snv) == util.golden_text(case, 'snv')
    assert util.frame_text(df_insdel) == util.golden_text(case, 'insdel')


def test_config1_one_megabase_contig_vs_reference_digest(built):
    """BASELINE.json configs[0] (one 1 Mb contig vs a 1 Mb chr20 slice, the reference's own CPU-runnable case): the oracle's
    tables, with the FILTER of rule call_cigar, as text == what pavlib.cigarcall wrote for the same seeded input in the build
    container (md5 committed by tools/refharness/gen_golden_cigar.py; the 2 MB of sequence are regenerated from the seed)."""
    from pav_amd import cigarcall, rules
    hap, gold = util.config1_case()
    names = hap.ref.names
    snv, indel, blob, err = util.oracle_records(names, [hap.ref.seqs[n] for n in names], hap.tig_names,
                                                [hap.tig_seqs[n] for n in hap.tig_names], hap.df_align)
    assert err.kind == 0
    df_snv, df_insdel = cigarcall.records_to_frames(snv, indel, blob, hap.df_align, 'h1')
    util.assert_config1_tables(rules.apply_trim_filter(df_snv, hap.df_trim), rules.apply_trim_filter(df_insdel, hap.df_trim), gold)


def test_empty_table(built):
    d, df_align, df_trim = util.golden_case('cigar_empty')
    df_snv, df_insdel = util.oracle_frames(d, df_align, df_trim, with_filter=False)
    assert util.frame_text(df_snv) == util.golden_text('cigar_empty', 'snv')
    assert util.frame_text(df_insdel) == util.golden_text('cigar_empty', 'insdel')


def test_error_cases(built):
    """Same exception type and message as the reference (tests/golden/cigar_errors.json)."""
    from pav_amd import cigarcall
    d, df_align, _ = util.golden_case('cigar_edge')
    ref_fa, tig_fa = util.seq_arrays(d, df_align)
    for e in util.cigar_errors():
        df = df_align.copy()
        for idx, cig in e['edits'].items():
            df.loc[df['INDEX'] == int(idx), 'CIGAR'] = cig
        _, _, _, err = util.oracle_records(ref_fa.names, [ref_fa[n] for n in ref_fa.names], tig_fa.names,
                                           [tig_fa[n] for n in tig_fa.names], df)
        assert err.kind != 0, e['label']
        with pytest.raises((RuntimeError, IndexError)) as ei:
            cigarcall._raise_reference_error(err, df)
        assert type(ei.value).__name__ == e['type'], e['label']
        assert str(ei.value) == e['message'], e['label']


def test_rule_outputs_equal_the_reference_rules(built):
    """Oracle tables, split into the 10 CALL_BATCH groups and merged like rule call_cigar_merge, against the text the
    reference's own rule bodies wrote (tests/golden/rule_call_cigar): byte-identical."""
    import io
    import pandas as pd
    d, df_align, df_trim = util.golden_case('cigar_synth')
    parts_snv, parts_ins = [], []
    for batch in range(10):
        sub = df_align.loc[df_align['CALL_BATCH'] == batch]
        df_snv, df_insdel = util.oracle_frames(d, sub, df_trim)
        # the rule writes each batch to disk and the merge rule reads it back (dtypes are re-inferred by read_csv)
        for frames, part in ((parts_snv, df_snv), (parts_ins, df_insdel)):
            frames.append(pd.read_csv(io.StringIO(util.frame_text(part)), sep='\t', keep_default_na=False))
    snv = pd.concat(parts_snv, axis=0).reset_index(drop=True).sort_values(['#CHROM', 'POS'])
    ins = pd.concat(parts_ins, axis=0).reset_index(drop=True).sort_values(['#CHROM', 'POS', 'END', 'ID'])
    assert util.frame_text(ins) == util.golden_text('rule_call_cigar', 'insdel_merged')
    assert util.frame_text(snv) == util.golden_text('rule_call_cigar', 'snv_merged')
