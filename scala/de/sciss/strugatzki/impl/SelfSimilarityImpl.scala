/*
 *  SelfSimilarityImpl.scala -- replacement of the reference's class of this name.  The gray colour scheme is
 *  rendered and written as PNG by the library.  The psycho-optical scheme (the default) uses a palette class of
 *  the JVM side (de.sciss.intensitypalette.IntensityPalette): for it the library hands over the similarity
 *  matrix in bands of rows -- it reads the inputs and works out span, window and decimation itself -- and only
 *  the colouring of a pixel and its mirror image happens here (the reference's lines 150-155, 167).
 *  Parity of the palette: unpinned (third party).
 */
package de.sciss.strugatzki
package impl

import java.awt.image.{BufferedImage, DataBufferInt}
import java.nio.{ByteBuffer, ByteOrder}
import javax.imageio.ImageIO

import de.sciss.intensitypalette.IntensityPalette
import de.sciss.processor.impl.ProcessorImpl

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

  private def psycho(ctx: Long): Unit = {
    val xml    = config.toXML.toString
    val imgExt = Native.selfSimilarityRows(ctx, xml, 0, 0, null)
    // a band of rows per call: at most 256 MB of similarities (a direct buffer ends at 2 GB; the reference's
    // largest raster, 46340 x 46340, is 8.6 GB of them)
    val bandRows = math.max(1, math.min(imgExt.toLong, (64L << 20) / math.max(1, imgExt))).toInt
    val sims = ByteBuffer.allocateDirect(bandRows * imgExt * 4).order(ByteOrder.nativeOrder()).asFloatBuffer()
    val colorScale = 1.0f / config.colorCeil
    val img     = new BufferedImage(imgExt, imgExt, BufferedImage.TYPE_INT_RGB)
    val imgData = img.getRaster.getDataBuffer.asInstanceOf[DataBufferInt].getData()
    var x0 = 0
    while (x0 < imgExt) {
      checkAborted()
      val x1 = math.min(imgExt, x0 + bandRows)
      Native.selfSimilarityRows(ctx, xml, x0, x1, sims)
      var x = x0
      while (x < x1) {
        val row = (x - x0) * imgExt
        var y = x
        while (y < imgExt) {
          val v    = math.pow(math.max(0f, sims.get(row + y)), config.colorWarp).toFloat * colorScale
          val colr = IntensityPalette.apply(if (config.colorInv) 1f - v else v)
          imgData((imgExt - 1 - y) * imgExt + x) = colr
          imgData((imgExt - 1 - x) * imgExt + y) = colr
          y += 1
        }
        x += 1
      }
      x0 = x1
      progress = 0.9 * x0 / imgExt
    }
    ImageIO.write(img, "png", config.imageOutput)
    progress = 1.0
  }
}
