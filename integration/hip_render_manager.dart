/****************************************************************************
 * hip_render_manager.dart -- RenderManager's fan-out over the GPUs of one node.
 *
 * The reference spreads a render over `numThreads` isolates, each loading the scene itself, rendering the
 * GetSubWindow rectangle of its taskNum and posting an OutputImage that the manager copies into place
 * (lib/dartray_web/render_manager.dart:100-141; lib/dartray_web/render_isolate.dart:31-41;
 * lib/core/common.dart:52-73).  The device library keeps one GPU context and one RCCL communicator per PROCESS,
 * so the unit here is a process, not an isolate:
 *
 *   manager (this file, `HipRenderManager.render`)
 *     |- starts N workers:  dart hip_render_manager.dart --worker <scene.pbrt> <rank> <N> <dir>
 *     |     worker r: dr_init(r); every rank says whether it can join (commAvailable) and all agree through files in <dir>;
 *     |     rank 0 draws the RCCL unique id (HipSamplerRenderer.commUniqueId), writes it to <dir>/id, all agree again;
 *     |     the others read the id; all join the communicator (commInit);
 *     |     every worker loads the scene with the 'hipsampler' renderer, tile share r of N (round-robin 32 x 32
 *     |     tiles: balanced even when the geometry sits in one corner of the image);
 *     |     HipSamplerRenderer.render -> dr_render_sharded: the tiles, ONE ncclReduce(sum, f32) of the full-frame
 *     |     (X, Y, Z, weight) film over xGMI onto rank 0, ImageFilm.writeImage there;
 *     |     rank 0 writes the OutputImage to <dir>/out.f32 (width, height, then rgb)
 *     `- reads <dir>/out.f32 and completes with the OutputImage -- no rectangle copies: the reduce is the merge, and
 *        it also carries the filter splats across tile borders that the reference's copy drops.
 *
 * Not executed here (no Dart SDK in the image this was written in); the same sequence runs from Python
 * (dartray_amd/dist.py, bench.py --gpus N, tests/test_gpu_comm.py) and the entry points it binds are checked
 * against include/dartray_hip.h by tests/test_abi_c_host.py.  SDK window: see hip_sampler_renderer.dart.
 ****************************************************************************/
library hip_render_manager;

import 'dart:async';
import 'dart:io';
import 'dart:typed_data';

import '../core/core.dart';
import '../dartray/dartray.dart';
import 'hip_sampler_renderer.dart';

class HipRenderManager {
  /// Renders [scenePath] on [numGpus] GPUs of this node; completes with rank 0's OutputImage.
  Future<OutputImage> render(String scenePath, int numGpus) async {
    if (numGpus <= 1) {
      // one GPU: the plain path -- DartRay.loadScene with the 'hipsampler' renderer (INTEGRATION.md section 2)
      return await _renderInProcess(scenePath, 0, 1);
    }
    Directory dir = await Directory.systemTemp.createTemp('dartray_hip_');
    try {
      List<Future<int>> exits = [];
      for (int rank = 0; rank < numGpus; ++rank) {
        Process p = await Process.start(Platform.resolvedExecutable,
            [Platform.script.toFilePath(), '--worker', scenePath, '$rank', '$numGpus', dir.path],
            environment: {'HSA_ENABLE_IPC_MODE_LEGACY': '0'});  // dmabuf IPC between the ranks' GPU buffers
        stdout.addStream(p.stdout);
        stderr.addStream(p.stderr);
        exits.add(p.exitCode);
      }
      List<int> codes = await Future.wait(exits);
      for (int rank = 0; rank < numGpus; ++rank) {
        if (codes[rank] != 0) {
          LogSevere('HipRenderManager: worker $rank exited with ${codes[rank]}');
        }
      }
      return _readOutput(new File('${dir.path}/out.f32'));
    } finally {
      await dir.delete(recursive: true);
    }
  }

  // ---- worker side ----
  static Future<void> worker(String scenePath, int rank, int world, String dirPath) async {
    // The handshake is symmetric phase by phase (as dartray_amd/dist.py's): every phase ends with a file per rank that says
    // "ok" or "fail", and nobody enters the next phase -- in particular nobody blocks in ncclCommInitRank -- unless every
    // rank said ok.  Phase 1: can this process join at all (dr_comm_available: binds librccl, talks to nobody)?
    bool ok = true;
    try {
      ok = HipSamplerRenderer.commAvailable(rank);
    } catch (e) {
      ok = false;
    }
    await _agree(dirPath, 'available', rank, world, ok);
    // Phase 2: rank 0 draws the unique id.
    File idFile = new File('$dirPath/id');
    Uint8List id;
    if (rank == 0) {
      try {
        id = HipSamplerRenderer.commUniqueId(rank);
        File tmp = new File('$dirPath/id.tmp');
        await tmp.writeAsBytes(id, flush: true);
        await tmp.rename(idFile.path);  // atomic: the others never see a half-written id
      } catch (e) {
        ok = false;
      }
    }
    await _agree(dirPath, 'id', rank, world, ok);
    if (rank != 0) {
      id = await idFile.readAsBytes();
    }
    HipSamplerRenderer.commInit(rank, rank, world, id);   // device == rank: one GPU per process
    try {
      OutputImage out = await _renderInProcess(scenePath, rank, world);
      if (rank == 0) {
        await _writeOutput(new File('$dirPath/out.f32'), out);
      }
    } finally {
      HipSamplerRenderer.commDestroy();
    }
  }

  /// One phase of the workers' handshake: publish this rank's verdict, wait (bounded) for every rank's, and leave the job -- all
  /// ranks alike, exit code 3 -- if anyone failed or never answered.
  static Future<void> _agree(String dirPath, String phase, int rank, int world, bool ok) async {
    File mine = new File('$dirPath/$phase.$rank.tmp');
    await mine.writeAsString(ok ? 'ok' : 'fail', flush: true);
    await mine.rename('$dirPath/$phase.$rank');
    DateTime deadline = new DateTime.now().add(const Duration(seconds: 120));
    for (int r = 0; r < world; ++r) {
      File f = new File('$dirPath/$phase.$r');
      while (!await f.exists()) {
        if (new DateTime.now().isAfter(deadline)) {
          LogSevere('HipRenderManager: rank $r never reached phase "$phase"');
          exit(3);
        }
        await new Future.delayed(const Duration(milliseconds: 5));
      }
      if ((await f.readAsString()) != 'ok') {
        LogSevere('HipRenderManager: rank $r failed in phase "$phase"');
        exit(3);
      }
    }
  }

  /// DartRay.loadScene with the renderer overridden: 'hipsampler' with this process's tile share
  /// (the `hipsampler` branch of DartRay._makeRenderer reads 'tilerank' / 'tilecount' / 'device' from the ParamSet,
  /// INTEGRATION.md section 2).
  static Future<OutputImage> _renderInProcess(String scenePath, int rank, int world) {
    // RenderOverrides.rendererName / rendererParams replace the scene file's Renderer directive
    // (dartray.dart:62-70,665-668; core/render_overrides.dart:41-42)
    RenderOverrides overrides = new RenderOverrides();
    overrides.rendererName = 'hipsampler';
    overrides.rendererParams.addInt('tilerank', [rank]);
    overrides.rendererParams.addInt('tilecount', [world]);
    overrides.rendererParams.addInt('device', [rank]);
    DartRay dartray = new DartRay(new ResourceManager());
    return dartray.renderScene(scenePath, overrides: overrides);
  }

  static Future<void> _writeOutput(File f, OutputImage out) async {
    ByteData head = new ByteData(8);
    head.setInt32(0, out.imageWidth, Endian.little);
    head.setInt32(4, out.imageHeight, Endian.little);
    IOSink s = f.openWrite();
    s.add(head.buffer.asUint8List());
    s.add(out.rgb.buffer.asUint8List(out.rgb.offsetInBytes, out.rgb.lengthInBytes));
    await s.close();
  }

  static Future<OutputImage> _readOutput(File f) async {
    Uint8List bytes = await f.readAsBytes();
    ByteData head = new ByteData.view(bytes.buffer, bytes.offsetInBytes, 8);
    int w = head.getInt32(0, Endian.little), h = head.getInt32(4, Endian.little);
    Float32List rgb = new Float32List.fromList(bytes.buffer.asFloat32List(bytes.offsetInBytes + 8, 3 * w * h));
    return new OutputImage(0, 0, w, h, w, h, rgb);
  }
}

/// `dart hip_render_manager.dart <scene.pbrt> <numGpus>` renders; `--worker ...` is what the manager starts.
Future<void> main(List<String> args) async {
  if (args.length == 5 && args[0] == '--worker') {
    await HipRenderManager.worker(args[1], int.parse(args[2]), int.parse(args[3]), args[4]);
    return;
  }
  OutputImage out = await new HipRenderManager().render(args[0], args.length > 1 ? int.parse(args[1]) : 1);
  LogInfo('rendered ${out.imageWidth} x ${out.imageHeight}');
}
