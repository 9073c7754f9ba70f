"""Stream ordering between the library's draws and a caller's own work (collectives, copies) on another stream."""
import pytest
import torch

import renderer_amd as R
from renderer_amd import dist as RD

pytestmark = pytest.mark.gpu


def test_current_stream_is_ordered_after_a_context_with_its_own_stream(bunny_path):
    """A context created without a stream draws on a private non-blocking stream.  What the composite enqueues on
    torch's current stream (the gather) must see the finished draw: order_current_stream_after makes it wait on the
    device.  Eight 8K draws are queued, so an unordered copy would run long before they end."""
    ctx = R.Context(0)
    assert ctx.stream and ctx.stream != torch.cuda.current_stream().cuda_stream
    scene = R.Scene(ctx, R.Mesh.load_obj(bunny_path))
    try:
        w, h = 7680, 4320
        buf = torch.zeros(3, w * h, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        cam, prm = R.camera_reference(w, h), R.make_params(w, h, flags=R.LAMBERT)
        for _ in range(8):
            scene.render_device(cam, prm, buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr())
        RD.order_current_stream_after(ctx)
        seen = buf.clone()  # torch's current stream: stands for the collective that reads the drawn tiles
        torch.cuda.current_stream().synchronize()
        hits = int((seen[2] != -1).sum())
        assert 390000 < hits < 410000  # 16 x the 1080p coverage (25 139), give or take edge pixels
        ctx.synchronize()
        assert torch.equal(seen, buf)
        # and the other way round: a draw enqueued after the caller's work waits for it
        buf.zero_()  # current stream
        RD.order_context_stream_after_current(ctx)
        scene.render_device(cam, prm, buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr())
        ctx.synchronize()
        assert torch.equal(seen, buf)
    finally:
        scene.close()
        ctx.close()


def test_context_reports_the_stream_it_was_given():
    s = torch.cuda.Stream()
    ctx = R.Context(0, stream=s.cuda_stream)
    assert ctx.stream == s.cuda_stream
    with torch.cuda.stream(s):
        RD.order_current_stream_after(ctx)  # one stream: nothing to order, nothing raised
    ctx.close()


def test_context_outlives_its_scenes(bunny_path):
    """inv_ctx_destroy refuses while a scene built on the context is alive (the scene holds a pointer to it)."""
    ctx = R.Context(0)
    scene = R.Scene(ctx, R.Mesh.load_obj(bunny_path))
    with pytest.raises(R.InvertError) as e:
        ctx.close()
    assert e.value.code == R.api.ERR_ARG
    scene.close()
    ctx.close()
