"""Host-side input / output pipeline (idm_vton_amd/io.py): rank-sharded iteration of the scripts' DataLoaders
and the asynchronous image writer."""
import os

import numpy as np
import pytest
import torch

from idm_vton_amd import io as pio


class _DS(torch.utils.data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {"idx": i, "x": torch.full((2,), float(i))}


@pytest.mark.parametrize("n,world", [(7, 2), (8, 8), (3, 4), (16, 8), (1, 2)])
def test_sharded_sampler_visits_every_image_exactly_once(n, world):
    seen = []
    for r in range(world):
        s = pio.ShardedSampler(n, r, world)
        idx = list(s)
        assert len(idx) == len(s) and all(s.global_index(p) == i for p, i in enumerate(idx))
        seen += idx
    assert sorted(seen) == list(range(n))
    sizes = [len(pio.ShardedSampler(n, r, world)) for r in range(world)]
    assert max(sizes) - min(sizes) <= 1


def test_shard_dataloaders_patches_the_unsharded_loader_a_script_builds():
    undo = pio.shard_dataloaders(1, 2)
    try:
        dl = torch.utils.data.DataLoader(_DS(7), shuffle=False, batch_size=2, num_workers=0)     # inference.py:309-314's form
        got = [int(i) for b in dl for i in b["idx"]]
        assert got == [1, 3, 5]
        dl2 = torch.utils.data.DataLoader(_DS(7), shuffle=True, batch_size=2)                     # a shuffling loader is left alone
        assert len(dl2.sampler) == 7
    finally:
        undo()
    assert len(torch.utils.data.DataLoader(_DS(7), batch_size=1).sampler) == 7                    # undone
    assert pio.shard_dataloaders(0, 1)() is None                                                   # world 1: no-op


def test_async_writer_writes_the_same_bytes_as_the_synchronous_path(tmp_path):
    from PIL import Image
    g = torch.Generator().manual_seed(3)
    imgs = [torch.rand(3, 64, 48, generator=g) for _ in range(5)] + [torch.rand(1, 3, 32, 32, generator=g)]
    w = pio.AsyncImageWriter(workers=2, max_pending=2)
    for i, t in enumerate(imgs):
        w.submit(t, str(tmp_path / "a" / f"{i}.png"))
    w.close()
    for i, t in enumerate(imgs):
        t3 = t[0] if t.ndim == 4 else t
        a = t3.mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
        Image.fromarray(a).save(str(tmp_path / f"s{i}.png"))
        assert open(tmp_path / "a" / f"{i}.png", "rb").read() == open(tmp_path / f"s{i}.png", "rb").read()


def test_async_writer_reports_errors_on_flush(tmp_path):
    w = pio.AsyncImageWriter(workers=1)
    w.submit(torch.rand(3, 8, 8), str(tmp_path / "x.unknownext"))
    with pytest.raises(Exception):
        w.flush()
    w.submit(torch.rand(3, 8, 8), str(tmp_path / "ok.jpg"))
    w.close()
    assert os.path.getsize(tmp_path / "ok.jpg") > 0


def test_save_image_async_rejects_what_it_would_silently_get_wrong(tmp_path):
    """a batch of N > 1 images (torchvision writes a grid; dropping N - 1 of them silently is the worst outcome) and
    torchvision keywords that mean something else to PIL are errors, not guesses."""
    import pytest
    import torch
    from idm_vton_amd import io as pio
    with pytest.raises(ValueError, match="batch of 2"):
        pio.save_image_async(torch.rand(2, 3, 8, 8), str(tmp_path / "a.png"))
    with pytest.raises(TypeError, match="quality_level"):
        pio.save_image_async(torch.rand(3, 8, 8), str(tmp_path / "b.png"), quality_level=3)
    with pytest.raises(NotImplementedError, match="normalize"):
        pio.save_image_async(torch.rand(3, 8, 8), str(tmp_path / "c.png"), normalize=True)
    pio.save_image_async(torch.rand(1, 3, 8, 8), str(tmp_path / "d.png"), nrow=8, padding=2)     # neutral grid keywords: fine
    pio.default_writer().flush()
    assert (tmp_path / "d.png").exists()


def test_launcher_exits_nonzero_when_a_queued_image_cannot_be_written(tmp_path):
    """write errors of the asynchronous writer used to surface in an atexit hook (printed, exit status 0)."""
    import subprocess
    import sys
    script = tmp_path / "s.py"
    (tmp_path / "taken.png").mkdir()                      # the target exists as a DIRECTORY: the encode runs, the write fails on the worker thread
    script.write_text("import torch, torchvision\n"
                      f"torchvision.utils.save_image(torch.rand(3, 8, 8), {str(tmp_path / 'taken.png')!r})\nprint('script done')\n")
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    r = subprocess.run([sys.executable, __import__("os").path.join(root, "tests", "dropin_launcher.py"), str(script)], capture_output=True, text=True, timeout=300)
    assert "script done" in r.stdout and r.returncode != 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
