/*
 *  SelfSimilarityImpl.scala -- replacement of the reference's class of this name.  The gray colour scheme is
 *  rendered and written as PNG by the library.  The psycho-optical scheme (the default) uses a palette class of
 *  the JVM side (de.sciss.intensitypalette.IntensityPalette): for it the library returns the similarity matrix
 *  and the pixels are coloured here, exactly as the reference's loop does (its lines 98-115, 150-155, 167).
 */
package de.sciss.strugatzki
package impl

import java.awt.image.{BufferedImage, DataBufferInt}
import java.io.File
import java.nio.{ByteBuffer, ByteOrder, FloatBuffer}
import javax.imageio.ImageIO

import de.sciss.intensitypalette.IntensityPalette
import de.sciss.processor.impl.ProcessorImpl
import de.sciss.span.Span
import de.sciss.synth.io.AudioFile

import scala.concurrent.blocking

private[strugatzki] final class SelfSimilarityImpl(val config: SelfSimilarity.Config)
  extends SelfSimilarity with ProcessorImpl[SelfSimilarity.Product, SelfSimilarity] {

  import SelfSimilarity._

  protected def body(): Product = blocking {
    Native.withContexts(Native.devices.take(1)) { ctx =>
      if (config.colors == GrayScale)
        Native.run(this, p => { progress = p }, () => checkAborted())(Native.runSelfSimilarity(ctx(0), config.toXML.toString, _))
      else psycho(ctx(0))
    }
  }

  private def direct(n: Long): FloatBuffer =
    ByteBuffer.allocateDirect((n * 4).toInt).order(ByteOrder.nativeOrder()).asFloatBuffer()

  /** Interleaved frames of a feature file in a direct buffer (the layout the library takes). */
  private def load(f: File): (FloatBuffer, Long, Int) = {
    val af = AudioFile.openRead(f)
    try {
      val n   = af.numFrames
      val buf = af.buffer(8192)
      val out = direct(n * af.numChannels)
      var off = 0L
      while (off < n) {
        val len = math.min(8192L, n - off).toInt
        af.read(buf, 0, len)
        var i = 0
        while (i < len) { var ch = 0; while (ch < af.numChannels) { out.put(buf(ch)(i)); ch += 1 }; i += 1 }
        off += len
      }
      (out, n, af.numChannels)
    } finally af.close()
  }

  private def psycho(ctx: Long): Unit = {
    val extr1 = FeatureExtraction.Config.fromXMLFile(config.metaInput)
    val extr2 = config.metaInput2.fold(extr1)(FeatureExtraction.Config.fromXMLFile)
    require(extr1.fftSize == extr2.fftSize && extr1.fftOverlap == extr2.fftOverlap && extr1.numCoeffs == extr2.numCoeffs)
    val stepSize   = extr1.fftSize / extr1.fftOverlap
    def fullToFeat(n: Long): Int = ((n + (stepSize >> 1)) / stepSize).toInt
    val halfWinLen = fullToFeat(config.corrLen)
    val (f1, n1, numCh) = load(extr1.featureOutput)
    val same            = extr1.featureOutput == extr2.featureOutput
    val (f2, n2, _)     = if (same) (null, n1, numCh) else load(extr2.featureOutput)
    val norms = if (!config.normalize) null else {
      val af = AudioFile.openRead(new File(config.databaseFolder, Strugatzki.NormalizeName))
      try {
        require(af.numChannels == numCh && af.numFrames == 2L)
        val b = af.buffer(2); af.read(b)
        val d = direct(2L * numCh)
        var ch = 0; while (ch < numCh) { d.put(b(ch)(0)); d.put(b(ch)(1)); ch += 1 }
        d
      } finally af.close()
    }
    // geometry: the reference's lines 64-91
    val afNumFrames = math.min(n1, n2)
    val afStart = config.span match { case Span.HasStart(s) => math.max(0, fullToFeat(s)); case _ => 0 }
    val afStop  = config.span match { case Span.HasStop(s) => math.min(afNumFrames.toInt, fullToFeat(s)); case _ => afNumFrames.toInt }
    val numCorrs = math.max(0L, (afStop - afStart) - 2 * halfWinLen + 1).toInt
    require(config.decimation >= 1, "Illegal decimation setting of " + config.decimation)
    val (decim, imgExt) = {
      val i = numCorrs / config.decimation
      if (i <= 0xB504) (config.decimation, i) else { val d1 = (numCorrs + 0xB503) / 0xB504; (d1, numCorrs / d1) }
    }
    require(config.colorWarp > 0 && config.colorCeil > 0)
    val sims = direct(imgExt.toLong * imgExt)
    Native.selfSimilarityMatrix(ctx, f1, n1, f2, n2, numCh, norms, afStart, halfWinLen, decim, imgExt,
      config.temporalWeight, sims)
    progress = 0.9
    checkAborted()
    val colorScale = 1.0f / config.colorCeil
    val img     = new BufferedImage(imgExt, imgExt, BufferedImage.TYPE_INT_RGB)
    val imgData = img.getRaster.getDataBuffer.asInstanceOf[DataBufferInt].getData()
    var x = 0
    while (x < imgExt) {
      var y = x
      while (y < imgExt) {
        val sim  = sims.get(x * imgExt + y)
        val v    = math.pow(math.max(0f, sim), config.colorWarp).toFloat * colorScale
        val colr = IntensityPalette.apply(if (config.colorInv) 1f - v else v)
        imgData((imgExt - 1 - y) * imgExt + x) = colr
        imgData((imgExt - 1 - x) * imgExt + y) = colr
        y += 1
      }
      x += 1
    }
    ImageIO.write(img, "png", config.imageOutput)
    progress = 1.0
  }
}
