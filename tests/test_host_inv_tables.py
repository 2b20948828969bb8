"""pav_amd.rules.inv_batch_texts - the per-batch INV tables of rule call_inv_batch and the merged table of rule call_inv_batch_merge
written as text from the row values (what rules.call_haplotype does since the device writes the large tables and pandas on sixty
small frames became a third of the files-to-files time) - against the rules' own pandas code on the same rows: byte for byte,
including batches without regions (full header), batches with regions but no call (the header WITHOUT the FILTER column,
rules/call_inv.snakefile:300-308 sic - which moves FILTER behind SEQ in the merged table when such a batch comes first), duplicate
IDs across batches, ties on (#CHROM, POS).  Values pandas.read_csv would re-type on the way through the merge must make it refuse."""
import io

import numpy as np
import pandas as pd

from pav_amd import rules

COLS = rules.INV_BED_COLUMNS


def row(rng, i, chrom, pos):
    seq = ''.join(rng.choice(list('ACGTacgtN'), 50))
    return [chrom, np.int64(pos), pos + 100 + i, f'{chrom}-{pos + 1}-INV-{100 + i}', 'INV', np.int64(100 + i), 'h1', f'tig{i}:{pos}-{pos + 99}',
            '-' if i % 2 else '+', 0, f'{chrom}:{pos + 5}-{pos + 90}', f'tig{i}:{pos + 5}-{pos + 90}', f'{chrom}:{pos - 10}-{pos + 200}',
            f'tig{i}:{pos - 10}-{pos + 200}', f'{chrom}-{pos}-RGN-7', 'RGN', '12' if i % 3 else '12,15', 'INV_K', 'PASS', seq]


def pandas_texts(batch_rows):
    out = []
    for rows in batch_rows:
        if rows is None:
            df = pd.DataFrame([], columns=list(COLS))                                            # call_inv.snakefile:148-167
        elif rows:
            df = pd.concat([pd.Series(r, index=COLS) for r in rows], axis=1).T.sort_values(['#CHROM', 'POS', 'END', 'ID'])   # :297
        else:
            df = pd.DataFrame([], columns=[c for c in COLS if c != 'FILTER'])                     # :300-308
        out.append(df.to_csv(None, sep='\t', index=False))
    return out


def test_texts_equal_the_rules_pandas_code():
    rng = np.random.default_rng(0)
    for first_quirk in (False, True):
        batch_rows = []
        for b in range(16):
            if b % 4 == 0:
                batch_rows.append([] if (b == 0 and first_quirk) else None)
            elif b % 4 == 1:
                batch_rows.append([])
            else:
                batch_rows.append([row(rng, b * 10 + k, 'chr' + str(int(rng.integers(1, 12))), int(rng.integers(1, 10 ** 7))) for k in range(b % 3 + 1)])
        batch_rows[6] = [list(r) for r in batch_rows[2]]                                          # the same calls again: duplicate IDs
        batch_rows[7] = batch_rows[7] + [row(rng, 999, batch_rows[7][0][0], int(batch_rows[7][0][1]))]   # a tie on (#CHROM, POS)
        texts, merged, n = rules.inv_batch_texts(batch_rows)
        want = pandas_texts(batch_rows)
        assert texts == want
        df = rules.call_inv_batch_merge([io.BytesIO(t.encode()) for t in want], None)
        assert merged == df.to_csv(None, sep='\t', index=False) and n == df.shape[0]
        assert (df.columns.tolist()[-1] == 'FILTER') == first_quirk                               # the reference's column quirk, reproduced


def test_values_that_pandas_would_retype_are_refused():
    rng = np.random.default_rng(1)
    for col, value in ((0, '7'), (6, 'NA'), (16, '007'), (16, '1e5'), (6, ' h1'), (19, 'nan'), (15, 'True'), (6, 'a"b')):
        bad = [row(rng, 1, 'chr1', 100)]
        bad[0][col] = value
        assert rules.inv_batch_texts([bad]) is None, (col, value)
    assert rules.inv_batch_texts([[row(rng, 1, 'chr1', 100)]]) is not None
